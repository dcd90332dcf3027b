// Host stages of the pipeline (serial bit parsing / packing; no GPU involved).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <utility>
#include <vector>

#include "../../include/mp3s.h"
#include "mp3s_tables.h"

namespace mp3s {

struct ParsedStream {
    int n_frames = 0, nch = 0, sampling_rate = 0, bit_rate = 0, dup_last_frame = 0;
    std::vector<int16_t> is;              // [n][2 gr][2 ch][576]
    std::vector<mp3s_granule_si> si;      // [n][2][2]
    std::vector<mp3s_frame_hdr> hdr;      // [n]
    std::vector<uint8_t> bits;            // stego bits
    std::vector<int32_t> table_select;    // [n][2][2][3]
    std::vector<int32_t> frame_size;      // [n]
};
// byte-level scan only (no scalefactor / Huffman decode): what the device Huffman kernel consumes
// vectors whose resize() leaves new elements uninitialised: the scan sizes them to their capacity up front and writes
// every byte it later counts, so zero-filling tens of megabytes first would only touch the pages twice
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
struct ScannedStream {
    std::vector<mp3s_frame_side, NoInitAlloc<mp3s_frame_side>> side;    // [n]
    std::vector<uint8_t, NoInitAlloc<uint8_t>> blob;   // main data of all frames, 4-byte aligned, >= 8 zero bytes after each
    bool gpu_ok = true;                   // false: some granule inherits scalefactors from earlier frames: the device
                                          // kernel resolves that by walking back through the stream's side records, which
                                          // needs the WHOLE stream in the batch (a block of such a stream is parsed on the host)
    bool host_parsed = false;             // the pipelines' front end took the host parser for this stream (side / blob unused)
};
// scan == nullptr: full parse (is + si); otherwise is/si stay empty and *scan is filled instead
int parse_stream(const uint8_t *file, size_t len, ParsedStream &out, ScannedStream *scan = nullptr);
// The same scan writing where the caller says: growable arrays (what parse_stream does with a ScannedStream) or arrays
// of fixed capacity -- page-locked staging of the asynchronous pipeline, which the copy engine reads directly.  A fixed
// sink that runs out of room ends the scan with MP3S_E_NOMEM.
struct ScanSink {
    uint8_t *blob = nullptr; size_t blob_len = 0, blob_cap = 0;
    mp3s_frame_side *side = nullptr; size_t n_side = 0, side_cap = 0;
    mp3s_frame_hdr *hdr = nullptr;     // optional, side_cap entries: the frame headers go here instead of ParsedStream::hdr
    bool gpu_ok = true;
    bool lean = false;                 // the caller wants neither stego bits nor table indices nor frame sizes
    void *user = nullptr;
    bool (*grow)(ScanSink *, size_t blob_need, size_t side_need) = nullptr;   // null: fixed capacity
};
int parse_stream_sink(const uint8_t *file, size_t len, ParsedStream &out, ScanSink *sink);

// Index of a stream: one walk over its frames (headers, side info, reservoir pointers -- no main data copied, nothing
// stored per frame) that leaves a resume point every 256 frames.  With it a block of the stream is scanned on its own:
// parse_stream_range starts at the resume point in front of the block, so a rank of a sharded stream or a chunk of a
// streamed file pays for its own frames only (plus at most 255 skipped ones), not for the whole file again.
struct StreamIndex;
struct StreamIndexInfo {
    long n_frames = 0;
    int nch = 0, sampling_rate = 0, bit_rate = 0;   // rate / bitrate of the LAST header, as parse_stream reports them
    int dup_last_frame = 0;
    bool gpu_ok = true;
};
StreamIndex *index_stream(const uint8_t *file, size_t len, StreamIndexInfo *info, int *rc);
void index_free(StreamIndex *ix);
const StreamIndexInfo &index_info(const StreamIndex *ix);
// frames [first, first + count) (clipped to the stream) into the sink, exactly as parse_stream_sink would have stored them
// (out.bits = the stego bits of these frames; out.dup_last_frame is not set: the index knows it)
int parse_stream_range(const uint8_t *file, size_t len, const StreamIndex *ix, long first, long count, ParsedStream &out, ScanSink *sink);
int parse_stream_range(const uint8_t *file, size_t len, const StreamIndex *ix, long first, long count, ParsedStream &out, ScannedStream &scan);

// scalefactors + Huffman of ONE frame of a scanned stream (side record + its main data in the blob): the host's answer for
// a frame the device Huffman kernel flags; exact for gpu_ok streams (no frame inherits anything from another)
int parse_scanned_frame(const mp3s_frame_side &fs, const uint8_t *blob, int16_t *is2304, mp3s_granule_si *si4);

// samplerate / bitrate -> header indices and whole slots per frame; non-zero if unsupported
int stream_params(int samplerate, int bitrate_kbps, int *sri, int *bri, int *whole_slots);

// per-frame padding bit and rate-loop budget (reference MP3_Encoder.py:503-513, 630-636, 894-912)
// frames [first_frame, first_frame + n_frames) of a stream; bytes_before = size of the frames in front of them
int rate_frames(int samplerate, int bitrate_kbps, int nch, int n_frames, mp3s_rate_frame *out, int32_t *padding,
                int64_t first_frame = 0, int64_t *bytes_before = nullptr);

// scfsi decision from the per-unit band energies (reference MP3_Encoder.py:861-892); en = [units][22]
void decide_scfsi(int n_frames, const int32_t *en, const mp3s_gr_out *gr, int32_t *scfsi /*[n][2][4]*/);

// __calc_scfsi energies of one granule*channel on the host with glibc log (reference MP3_Encoder.py:835-857);
// the device takes these from a table instead; kept as the table's cross-check (mp3s_debug_scfsi_energies, tests)
void host_scfsi_energies(const int32_t *xr576, int sr_idx, int32_t *en22);

// bitstream formatter (reference MP3_Encoder.py:1097-1145, 1266-1552); gr is modified (stuffing) on a copy
int format_stream(int samplerate, int bitrate_kbps, int n_frames, const int16_t *ix, const mp3s_gr_out *gr,
                  const int32_t *scfsi, std::vector<uint8_t> &mp3);

// ---- container formats and message framing (mp3s_host_files.cpp)
// reference encoder/WAV_Reader.py:30-111; *msg = the reference's sys.exit text when MP3S_E_EXIT is returned
int wav_parse(const uint8_t *file, size_t len, int bitrate_kbps, mp3s_wav_info *out, const char **msg);
// the 44-byte header scipy.io.wavfile.write emits for int16 data
void wav_header(int64_t n_rows, int nch, int rate, uint8_t *out44);
// reference steganography.py:10-24, 42-50
void message_frame(const uint8_t *utf8, size_t n, std::vector<uint8_t> &bits);
// reference decoder/decoder.py:90-108
void message_reveal(const uint8_t *bits, size_t n_bits, std::vector<uint8_t> &text);

}  // namespace mp3s
