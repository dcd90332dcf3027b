// Host-visible launchers of the device translation unit (mp3s_device.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mp3s.h"

namespace mp3s {

constexpr int DEC_SYNTH_TW = 4;   // waves per channel in a synthesis tile (tile = TW*64 - 15 output slots)

int dev_upload_tables(hipStream_t stream);

// scratch: H and TL planes, 2 * nch * 32 * (36 n + 18) doubles
size_t dec_scratch_bytes(int n_frames, int nch);
int launch_decode(hipStream_t stream, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr,
                  int n_frames, int nch, int n_halo, int out_format, void *d_pcm, void *d_scratch);

// scratch: subband samples int32 [2][32][36 n]
size_t enc_scratch_bytes(int n_frames);
int launch_encode(hipStream_t stream, const int16_t *d_pcm, const mp3s_frame_hdr *d_hdr, int n_frames, int32_t *d_mdct,
                  void *d_scratch);

// state: int32 [units][4] = address1, address2, address3, quantizerStepSize inherited from the previous frame
int launch_rate(hipStream_t stream, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                const uint8_t *d_hide, int n_hide, const int32_t *d_cursor, const int32_t *d_state,
                const int32_t *d_list, int n_list, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en);

}  // namespace mp3s
