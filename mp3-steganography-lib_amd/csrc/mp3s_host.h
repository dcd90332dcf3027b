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

// ---- the frame walk: all the host still does per frame when the side info is parsed and the main data gathered on the
//      device (launch_parse, k_parse.hpp).  The walk steps from header to header exactly as the reference's frame loop does
//      (decoder/MP3_Parser.py:68-80, Frame.py:288-316) and reads three things of every frame's side info: main_data_begin
//      (where the frame's main data starts: Frame.py:318-363), the part2_3_length fields (the longest granule sizes the
//      Huffman kernel's staging) and, for a re-encode, how many code books each granule uses (the cursor plan of the rate
//      loop).  It handles REGULAR streams -- MPEG-1 Layer III frames whose reservoir pointers stay inside the file -- and
//      says so when a stream is anything else (`irregular`): such a stream takes the byte-level scan above, which follows
//      the reference through every oddity.
using FrameRef = mp3s_frame_ref;     // one frame for the device parser, 16 bytes (include/mp3s.h)
using StreamRef = mp3s_stream_ref;   // one stream (or one chunk of a stream) of a batch, 40 bytes
struct WalkHeader {            // FrameHeader fields; they persist from frame to frame like the Python object's
    double version = 0; int layer = 0, crc = 0, bit_rate = 0, sampling_rate = 0, padding = 0, mode = 0, channels = 0;
    int mode_ext0 = 0; int sr_idx = -1;
};
struct FrameWalker {
    // 0, or the code parse_stream fails with before it reaches a frame
    int open(const uint8_t *file, size_t len);
    // up to `cap` more frames: refs[i] with file_off counted from the start of the file plus `image_base`, md_off running on
    // from md_cursor, stream = `stream`; tables4 (optional, [cap][4]): code books in use per granule in the ENCODER's unit
    // order (frame, channel, granule) for as long as `tables_wanted` says.  Returns the number of frames written; 0 once the
    // stream has ended (`ended`) or cannot be walked (`irregular` / `error`).
    long next(FrameRef *refs, long cap, uint8_t *tables4, uint32_t image_base, uint16_t stream);
    // Frame.__prev_frame_size as the gather of frame `f` sees it, from the refs of the stream's frames [0, f)
    static void history(const FrameRef *stream_refs, long f, uint16_t out[9]);
    // scalefactors + Huffman of the frame emitted LAST, on the host (the asynchronous paths decode a stream's last frame
    // here: SURVEY E14); *alone = nothing in it is inherited from another frame (otherwise the result must not be used)
    int decode_last(int16_t *is2304, mp3s_granule_si *si4, bool *alone);

    bool ended = false, dup_last = false, irregular = false;
    int error = 0;                       // the code parse_stream returns for this stream (set together with irregular)
    int nch = 0, sampling_rate = 0, bit_rate = 0;   // channel count of the first frame; rate / bitrate of the LAST header walked
    int max_p23 = 0;                     // largest part2_3_length so far
    bool any_silent = false;             // some granule without a code book in use so far
    long n_frames = 0;                   // frames emitted so far
    uint32_t md_cursor = 0;              // the next frame's md_off
    long tables_wanted = 0;              // tables4 is filled until this many code books have been counted (0: never)
    long tables_frames = 0, tables_seen = 0;   // frames whose tables4 entries are valid; code books counted in them

    // ---- the frame loop's state (MP3_Parser / Frame / FrameHeader objects)
    const uint8_t *file = nullptr; long flen = 0;
    WalkHeader hd;
    int frame_size = 0, prev[9] = {0}, first_nch = 0;
    long offset = 0;
    // ... and the same in front of the frame emitted last
    WalkHeader last_hd; int last_frame_size = 0, last_prev[9] = {0}; long last_offset = 0; int last_first_nch = 0;
    bool last_valid = false;             // ... (kept for the frames inside the stream's last 4 096 bytes only)
    // the header fields as the last full parse left them, and what it was a parse of
    uint32_t hd_key = 0; bool hd_key_ok = false; int fs_base = 0;
};
// stego bits of a stream from what the device parser leaves per frame (tsel[f]: the twelve table indices of the frame, five
// bits each in the order the reference walks them -- channel, granule, region -- and above them the four window-switching
// flags): a window-switching granule parses two regions only and keeps the third index of the frame before,
// which is serial and stays here.  `carry` = the four stale indices in front of the first frame (zeros at a stream's start),
// updated.  reference: decoder/Frame.py:676-685, decoder/util.py:67-81
void stego_bits_from_tsel(const uint64_t *tsel, long n_frames, int nch, uint8_t carry[4], std::vector<uint8_t> &bits);

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
