/*
 * mp3s.h -- C-ABI of the MI355X-native MP3 steganography hot path.
 *
 * Drop-in boundary for tomershay100/mp3-steganography-lib (mp3stego-lib 1.1.8).
 * The reference has no FFI layer: its boundary is the Python API re-exported at
 * reference mp3stego/__init__.py:1-4 (Decoder, Encoder, Steganography).  The
 * Python package shipped in mp3-steganography-lib_amd/mp3stego/ keeps that
 * surface and binds the entry points below with ctypes.
 * "replaces:" lines name the reference code each entry point stands in for.
 *
 * Conventions
 *   - every function returns MP3S_OK (0) or a negative MP3S_E_* code, never
 *     exits; mp3s_last_error() gives a message for the calling thread;
 *   - plain pointers + sizes, no C++/torch types; the caller owns all host
 *     buffers it passes in; buffers returned through mp3s_buf are owned by the
 *     library until mp3s_buf_free();
 *   - a context is bound to ONE HIP device (one process per GPU); calls on one
 *     context are serialised by the caller;
 *   - transforms run ONLY on the GPU: without a usable HIP device
 *     mp3s_ctx_create() fails with MP3S_E_NO_DEVICE and nothing falls back to
 *     the CPU.  The serial bit parsing / packing stages stay on the host by
 *     design.
 */
#ifndef MP3S_H
#define MP3S_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP3S_OK 0
#define MP3S_E_NO_DEVICE (-1)   /* no HIP device / HIP runtime error at init */
#define MP3S_E_HIP (-2)         /* HIP runtime error during a call */
#define MP3S_E_ARG (-3)         /* bad argument (null, size, range) */
#define MP3S_E_MALFORMED (-4)   /* input the reference raises IndexError/KeyError on */
#define MP3S_E_UNSUPPORTED (-5) /* input the reference itself cannot process (mono encode, partial frame) */
#define MP3S_E_STEP_RANGE (-6)  /* quantizer step left the table (reference: IndexError) */
#define MP3S_E_NOMEM (-7)
#define MP3S_E_EXIT (-8)        /* the reference calls sys.exit(text) here; the text is in mp3s_last_error() */
#define MP3S_E_BUSY (-9)        /* mp3s_pipe_submit: every slot is taken, collect a result first; mp3s_pipe_collect: nothing pending */
#define MP3S_E_TABLES (-10)     /* a constant table built with this host's libm is not the one a kernel was compiled for (the encoder's analysis
                                 * filter: csrc/analysis_plan.h); the encode entry points refuse, decoding works */

#define MP3S_PCM_I16 0 /* (pcm*32767) truncated toward zero, low 16 bits: reference decoder/MP3_Parser.py:91 */
#define MP3S_PCM_F32 1
#define MP3S_PCM_F64 2 /* the reference's own float64 samples, bit for bit */

typedef struct mp3s_ctx mp3s_ctx;
typedef struct mp3s_buf mp3s_buf; /* library-owned result, released with mp3s_buf_free() */

/* ---------------------------------------------------------------- records (device + host layout) */

/* per granule*channel side record consumed by the decode transform
 * replaces: the arguments of re_quantize(), reference decoder/Frame.py:157-176 */
typedef struct {
    uint8_t global_gain, scalefac_scale, block_type, mixed_block_flag, preflag;
    uint8_t sub_block_gain[3];
    uint8_t scale_fac_l[22];
    uint8_t scale_fac_s[3][13];
    uint8_t pad[3];
} mp3s_granule_si; /* 72 bytes */

/* per frame record.  stream_first = index (in the batch) of the frame that starts the
 * stream this frame belongs to: overlap/fifo state is zero before it
 * (reference decoder/Frame.py:234-235, encoder/MP3_Encoder.py:528-534). */
typedef struct {
    uint8_t sr_idx;    /* 0 = 44.1 kHz, 1 = 48 kHz, 2 = 32 kHz (header bits) */
    uint8_t nch;
    uint8_t ms_stereo; /* JointStereo && (byte3 & 0x20): reference Frame.py:273 */
    uint8_t flags;
    uint32_t stream_first;
} mp3s_frame_hdr; /* 8 bytes */

/* per frame input of the rate loop */
typedef struct {
    int32_t max_bits;     /* min(mean_bits // nch, 4095): reference MP3_Encoder.py:894-912 */
    int32_t sr_idx;
    int32_t hide_end;     /* the message of this frame's stream ends at hide[hide_end]: a batch of streams keeps their
                           * messages back to back in one array (mp3s_rate_frames sets INT32_MAX = the array's end) */
    int32_t stream;       /* index of the frame's stream in the batch (into mp3s_chain_seg[]); mp3s_rate_frames sets 0 */
} mp3s_rate_frame; /* 16 bytes */

/* per granule*channel result of the rate loop
 * replaces: GrInfo as left by __iteration_loop, reference encoder/MP3_Encoder.py:81-103, 760-815 */
typedef struct {
    int32_t part2_3_length; /* WITHOUT stuffing (added by the host back end, MP3_Encoder.py:1097-1145) */
    int32_t big_values, count1, quantizer_step, region0_count, region1_count, count1table_select;
    int32_t table_select[3];
    int32_t address[3];     /* address1/2/3 as left behind (persist per (gr,ch) across frames) */
    int32_t n_tables;       /* number of non-zero table_select: advance of the hide cursor (:808-809) */
    int32_t flags;          /* MP3S_RF_* */
    int32_t reserved0;      /* the address1/2/3 the unit was given, a1 | a2 << 10 | a3 << 20 (0 without d_state_in) */
    int32_t xrmax;
    int32_t reserved;
} mp3s_gr_out; /* 72 bytes */
#define MP3S_RF_ACTIVE 1      /* xrmax != 0: outer loop ran */
#define MP3S_RF_USED_ADDR_IN 2 /* the incoming address1/2/3 were read before being overwritten */
#define MP3S_RF_STEP_RANGE 4  /* quantizer step left the table */
#define MP3S_RF_LOG_GUARD 8   /* reserved: never set since the band energies come from a table built with the host's libm */
#define MP3S_RF_LISTED 16     /* the device's chain check has put the unit on its list of re-runs (mp3s_chain_redo_dev): the list entry's wave is the
                               * unit's one writer in that launch, a re-run that follows its chain of inheriting units stops in front of it */

/* ---------------------------------------------------------------- context */
int mp3s_ctx_create(int device, mp3s_ctx **out);
/* how many HIP devices this process can open (what a launcher of one process per GPU sizes itself by); 0 and MP3S_OK
 * on a host without one */
int mp3s_device_count(int *n);
void mp3s_ctx_destroy(mp3s_ctx *ctx);
const char *mp3s_last_error(void);
const char *mp3s_version(void);
int mp3s_device_name(mp3s_ctx *ctx, char *buf, size_t n);
/* the device's PCI address as sysfs spells it ("0000:c5:00.0"): what a monitor thread that must not touch the GPU reads clocks and
 * power by (/sys/bus/pci/devices/<address>/...; bench.py's `sustained` region) */
int mp3s_device_pci(mp3s_ctx *ctx, char *buf, size_t n);
int mp3s_sync(mp3s_ctx *ctx);
/* Stream order between two contexts of one device: work submitted to ctx after this call starts only when everything
 * submitted to other before it has finished.  This is how a second context runs the bit-level front end of the next
 * batch (mp3s_huffman_decode_dev) under the transform kernels of the current one. */
int mp3s_ctx_wait(mp3s_ctx *ctx, mp3s_ctx *other);
/* The same without a new record on other: work submitted to ctx after this call starts when what other's order event was LAST
 * recorded behind has finished -- the previous mp3s_ctx_wait(.., other), or, with MP3S_OPT_RATE_SIGNALS, other's last rate loop.
 * (Never recorded: no wait.)  Several contexts can wait for one point of another's stream for the price of one packet there, or none. */
int mp3s_ctx_wait_last(mp3s_ctx *ctx, mp3s_ctx *other);
/* Options of a context (what used to be MP3S_* environment switches read in the middle of a job: the environment now only
 * provides the DEFAULTS, read once by mp3s_ctx_create -- the names in brackets -- so two contexts of one process can differ
 * and no job path calls getenv).  Set them while nothing is in flight on the context. */
#define MP3S_OPT_SELECT 1          /* 1: the message cursor is decided on the device (mp3s_rate_select_dev); 0: guessed cursors,
                                    * checked and resolved afterwards [MP3S_NO_SELECT=1 -> 0] */
#define MP3S_OPT_REDO 2            /* 1: the chain check's re-runs on the device (mp3s_chain_redo_dev) [MP3S_NO_REDO=1 -> 0] */
#define MP3S_OPT_FAST_IMDCT 3      /* 1: int16 decode through the mirrored, fused IMDCT behind the guard [MP3S_FAST_IMDCT=0 -> 0] */
#define MP3S_OPT_PIPE_TAIL 4       /* a pipe created on this context (mp3s_pipe_create; not the context's own, which runs the chunks of one file and takes
                                    * none) puts a job's tail (selection, chain check, bit packing) on a stream of its own: 0 never, 1 (default) the
                                    * candidate its rehearsal likes best unless that clearly loses, 2 only if the rehearsal is faster with one [MP3S_PIPE_TAIL] */
#define MP3S_OPT_CHUNK_FRAMES 5    /* frames per chunk when ONE file goes through the overlapped stages (mp3s_hide_message, mp3s_clear_file,
                                    * mp3s_decode_file, mp3s_decode_stream); 0 = chosen from the file's length [MP3S_CHUNK_FRAMES] */
#define MP3S_OPT_DEVICE_PARSE 6    /* 1: side info and main-data gather on the device (mp3s_parse_frames_dev) wherever the stream is
                                    * regular; 0: the host's byte-level scan everywhere [MP3S_DEVICE_PARSE=0 -> 0] */
#define MP3S_OPT_FILE_PIPELINE 7   /* 1: the one-file calls above run as chunks through overlapped stages; 0: scan, upload, kernels,
                                    * download one after the other as in round 2 [MP3S_FILE_PIPELINE=0 -> 0] */
#define MP3S_OPT_SCAN_THREADS 8    /* host threads a multi-file call may use for its front end; 0 = from the CPUs this process may
                                    * run on (sched_getaffinity, cgroup quota) divided by the ranks on this host [MP3S_SCAN_THREADS] */
#define MP3S_OPT_FIRST_CHUNK_FRAMES 9 /* frames of the first chunk of a one-file call (the device starts when it has been walked); 0 = chosen
                                       * from the file's length; never shorter than what the message can reach [MP3S_FIRST_CHUNK_FRAMES] */
#define MP3S_OPT_FILE_UP 10        /* 1: a one-file call uploads the whole file from the context's helper thread; 0: each chunk's piece from the
                                    * calling thread, as files above 1 GB go [MP3S_NO_FILE_UP=1 -> 0] */
#define MP3S_OPT_HUF_LANES 11      /* decoding lanes per wave of the Huffman kernel: 0 = picked from the launch's size (32 up to 8 192 frames,
                                    * 64 beyond), or 8 | 16 | 32 | 62 (64 lanes in workgroups of two waves) | 64 [MP3S_HUF_LANES] */
#define MP3S_OPT_NUMA 12           /* 1: a pipe's workers and page-locked staging are bound to the CPUs of the GPU's NUMA node [MP3S_NO_NUMA=1 -> 0] */
#define MP3S_OPT_FLOAT_FAST 13     /* 1: float32 output may take the mirrored, fused IMDCT and the split synthesis of the int16 path (no guard:
                                    * the error is below 1e-9 of full scale, the contract's tolerance is 1e-5 relative); 0 (default): the float
                                    * formats are bit-identical to the reference [MP3S_FLOAT_FAST=1 -> 1] */
#define MP3S_OPT_FAIL_CHUNK 14     /* test aid: the k-th chunk (k = value, counted from 1) of the next one-file call fails with MP3S_E_HIP after
                                    * its front end has been queued; the option clears itself when it fires */
#define MP3S_OPT_FUSED_DECODE 15   /* 1 (default): the fast decode paths (int16; float32 under MP3S_OPT_FLOAT_FAST) run requantisation .. synthesis as ONE
                                    * kernel, a wave-local stream with no array between IMDCT and synthesis (k_decode_stream.hpp; timed as dec_synth);
                                    * 0: two kernels with the subband samples in device memory [MP3S_FUSED_DECODE=0 -> 0] */
#define MP3S_OPT_FUSED_ENCODE 16   /* 1: analysis filter bank and MDCT as ONE kernel, the subband samples between them in LDS (k_enc_fused; timed as enc_analysis,
                                    * no encode scratch, 221 MB less traffic per 10 000 frames); 0 (default): two kernels with the samples in device memory --
                                    * measured faster: the one kernel's workgroup barriers and its 67 KB of LDS per workgroup cost more than the round trip
                                    * through memory [MP3S_FUSED_ENCODE=1 -> 1] */
#define MP3S_OPT_PIPE_DEC 17       /* 1: a pipe created on this context runs the decode transform of job k + 1 on a stream of its own, under the encode
                                    * transforms and the rate loop of job k [MP3S_PIPE_DEC=0|1] */
#define MP3S_OPT_RATE_SIGNALS 18   /* 1: the dispatch of the rate loop (mp3s_rate_variants_dev, mp3s_rate_loop_dev) carries the context's order event as its
                                    * completion signal: another context's mp3s_ctx_wait_last(other, this) then waits for the rate loop without a
                                    * record packet of its own in this context's queue (a record or a wait between two kernels costs the stream 7-8 us
                                    * of nothing: tools/timeline.sh) [MP3S_RATE_SIGNALS=0|1, default 0] */
#define MP3S_OPT_PIPE_SIGNALS 19   /* bit 0: in a pipe and in the one-file calls the last decode dispatch carries the event the front end waits for as its own
                                    * completion signal, bit 1: the rate loop the event the tail stream waits for; 0 (default): event records behind them.
                                    * Measured (round 5, docs/LOG.md): 3 gives the pipe +1.5 % (18.0 -> 18.3 M frames/s) and now and then puts a context's
                                    * one-file calls into a slower order of its streams (1.1 -> 1.4 ms per 10 000 frames): off [MP3S_PIPE_SIGNALS=0..3] */
#define MP3S_OPT_COUNT 20
/* what became of the one-file calls of this context (mp3s_hide_message, mp3s_clear_file, mp3s_decode_file, mp3s_decode_stream,
 * mp3s_hide_message_chunked): files that went through the overlapped stages as chunks, their chunks, chunks that were run
 * again because they depended on a carry the guess got wrong, chunks whose chains the host resolved, and files that took
 * the stages one after the other instead (streams the frame walk does not take, mono re-encodes, ...) */
typedef struct {
    int64_t files, chunks, reruns, resolved, fallbacks;
    /* the context's own pipe (made by its first one-file call): what choosing its streams took (microseconds; part of that first
     * call's time), the miniature jobs rehearsed, the choice and whether stages seem to share hardware queues -- mp3s_pipe_stats */
    int64_t rehearsal_us, rehearsals, lanes, queue_shared;
} mp3s_run_stats;
int mp3s_ctx_run_stats(mp3s_ctx *ctx, mp3s_run_stats *out);
/* What a rank of a multi-process launch holds of the host (all optional): page-locked bytes this PROCESS keeps pooled between calls and the
 * cap it keeps them under (the ranks of a host share its lockable memory: 4 GB / LOCAL_WORLD_SIZE, at least 1 GB), the ranks it believes
 * share the host, the CPUs it may run on, and how many of those lie on the NUMA node of the context's GPU (what a pipe's workers and
 * staging are bound to with MP3S_OPT_NUMA; 0 = unknown, nothing is bound).  ctx may be NULL: the host's side alone (gpu_node_cpus = 0),
 * what a rank would be told without a GPU in reach (the eight-rank CPU test, tests/test_distributed.py). */
typedef struct {
    uint64_t pinned_pooled_bytes, pinned_pool_cap_bytes;
    int32_t local_world_size, cpus_allowed, gpu_node_cpus, reserved;
} mp3s_host_share;
int mp3s_ctx_host_share(mp3s_ctx *ctx, mp3s_host_share *out);
int mp3s_ctx_set_option(mp3s_ctx *ctx, int option, int64_t value);
int mp3s_ctx_get_option(mp3s_ctx *ctx, int option, int64_t *value);
/* host copy of the constant tables uploaded to the device (struct DevTables of csrc/mp3s_tables.h), for tests */
const void *mp3s_debug_tables(size_t *bytes);
/* host (glibc) evaluation of the __calc_scfsi energies of one granule*channel: en[0..20] bands, en[21] total.
 * The kernel's tabulated energies are cross-checked against this in the tests. */
int mp3s_debug_scfsi_energies(const int32_t *xr576, int sr_idx, int32_t *en22);

/* A probe of the int16 decode's guard (reference decoder/MP3_Parser.py:91 truncates pcm * 32767, decoder/Frame.py:65-154 is the
 * order of the sums it truncates): while d_x / d_eps (device arrays of `capacity` doubles) are set, the fused int16 decode (k_dec_stream)
 * also leaves per sample of a call its fast value x -- fp64, before the truncation -- and the width eps its guard compared with (infinite
 * for a granule it does not vouch for at all), at the sample's index in the call's PCM.  |x - 32767 * exact fp64 PCM| / eps is the margin
 * of the bound (tests/test_guard_margin.py asserts <= 0.5).  NULL, NULL, 0 ends the probe.  Not for production calls: 16 bytes more per sample.
 * The probe is an instantiation of the stream kernel: while it is set, an int16 decode that would not run that kernel (MP3S_OPT_FUSED_DECODE or
 * MP3S_OPT_FAST_IMDCT off, mp3s_synth_mode with scale 0) returns MP3S_E_ARG instead of leaving the arrays as they were; float formats never fill them. */
int mp3s_debug_guard_margin(mp3s_ctx *ctx, double *d_x, double *d_eps, int64_t capacity);

/* host decode (scalefactors + Huffman) of ONE frame of a scanned stream: what the stream pipelines do with the frames the
 * device Huffman kernel flags (exact for streams the scan marks gpu_ok); exposed for the tests.  side = that frame's
 * record, blob = the stream's blob; is2304 = int16 [2][2][576], si4 = mp3s_granule_si [2][2] */
int mp3s_debug_parse_scanned_frame(const void *frame_side, const uint8_t *blob, int16_t *is2304, mp3s_granule_si *si4);

/* the host's share of a job of the overlapped stages, alone: the frame walk of `file` over and over for about `seconds`
 * (the frame table written into one reused buffer, table counts for the first 1 000 code books as a hide job asks for them);
 * *frames_per_s = what one thread sustains.  No device involved (round 4 ran it in N processes side by side: docs/LOG.md). */
int mp3s_debug_walk_rate(const uint8_t *file, size_t len, double seconds, double *frames_per_s, int64_t *frames_per_pass);

/* device memory owned by the caller through the context (for resident pipelines / benchmarks) */
int mp3s_dev_alloc(mp3s_ctx *ctx, size_t bytes, void **dptr);
int mp3s_dev_free(mp3s_ctx *ctx, void *dptr);
int mp3s_dev_upload(mp3s_ctx *ctx, void *dptr, const void *host, size_t bytes);
int mp3s_dev_download(mp3s_ctx *ctx, void *host, const void *dptr, size_t bytes);
int mp3s_dev_memset(mp3s_ctx *ctx, void *dptr, int value, size_t bytes);
/* device to device on the context's stream, asynchronous (e.g. a template of assumed cursors into place at the top of a step) */
int mp3s_dev_copy(mp3s_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* HIP-event timer on the context's stream (the stream every kernel below is launched on) */
int mp3s_timer_start(mp3s_ctx *ctx);
int mp3s_timer_stop(mp3s_ctx *ctx, float *ms);
/* The int16 decode format takes FAST kernels -- an IMDCT of 18 sums and their mirror images, a synthesis by even / odd
 * splitting of the 32-point cosine sums (a sixth of the multiplications), both with fused multiply-adds -- behind a guard: a
 * sample whose value * 32767 lies within a proven bound of an integer is computed again from the Huffman output in the
 * reference's operation order (a fix-up kernel behind the synthesis), so the int16 PCM is the reference's, sample for sample
 * (derivation of the bound: DESIGN.md section 2; the float formats always run the exact kernels and are bit-identical to
 * the reference).
 * eps_scale: 1 = the proven bound (default), 0 = always the exact kernels, > 1 = a wider guard (tests: forces samples
 * through the exact path).  *exact_samples (optional) = samples the guard has sent there since the last call. */
int mp3s_synth_mode(mp3s_ctx *ctx, double eps_scale, int64_t *exact_samples);
/* a plain device-to-device copy kernel over `bytes` (read + write counted), `iters` launches timed with HIP events on the
 * context's stream: the HBM bandwidth a streaming kernel achieves on this device, to set beside the data-sheet peak */
int mp3s_bench_copy(mp3s_ctx *ctx, size_t bytes, int iters, double *gb_per_s);
/* per-kernel HIP-event timing on the same stream: enable, run, then collect the summed milliseconds and launch
 * counts of the kernels in this order: dec_imdct, dec_synth, enc_analysis, enc_mdct, rate_loop, dec_huffman,
 * enc_pack, chain (the pair of mp3s_chain_resolve_dev counts as one) (n >= MP3S_N_KERNELS) */
#define MP3S_N_KERNELS 8
int mp3s_profile_enable(mp3s_ctx *ctx, int on);
/* An event pair costs a few microseconds of stream time per launch: restrict the timing to the kernels in `mask` (bit k =
 * kernel k in the order above; all by default).  Takes effect with the next mp3s_profile_enable(ctx, 1). */
int mp3s_profile_select(mp3s_ctx *ctx, unsigned mask);
int mp3s_profile_collect(mp3s_ctx *ctx, double *total_ms, int64_t *launches, int n);

/* ---------------------------------------------------------------- (ii) decode transform batch
 * replaces: re_quantize, __ms_stereo, __reorder, __alias_reduction, imdct, __frequency_inversion,
 *           synth_filter_bank, __interleave -- reference decoder/Frame.py:65-218, 561-640 (and the
 *           (pcm*32767).astype(int16) of MP3_Parser.py:91 for MP3S_PCM_I16).
 * is  : int16 [n_frames][2 gr][2 ch][576]   (ch 1 ignored when nch == 1)
 * si  : mp3s_granule_si [n_frames][2][2]
 * hdr : mp3s_frame_hdr [n_frames]
 * pcm : [n_frames - n_halo][1152][nch] of the requested format
 * Frames are consecutive frames of one or more streams; the first n_halo frames only rebuild state
 * (a 1-frame halo is enough).  *_dev takes device pointers and is asynchronous on
 * the context's stream; the host variant uploads, runs, downloads and synchronises. */
int mp3s_decode_transform_dev(mp3s_ctx *ctx, const int16_t *d_is, const mp3s_granule_si *d_si,
                              const mp3s_frame_hdr *d_hdr, int n_frames, int nch, int n_halo, int out_format,
                              void *d_pcm);
int mp3s_decode_transform(mp3s_ctx *ctx, const int16_t *is, const mp3s_granule_si *si, const mp3s_frame_hdr *hdr,
                          int n_frames, int nch, int n_halo, int out_format, void *pcm);

/* ---------------------------------------------------------------- (iii) encode transform batch
 * replaces: __replace_samples, window_filter_sub_band, __mdct_sub (MDCT + alias butterflies)
 *           -- reference encoder/MP3_Encoder.py:321-370, 652-758.
 * pcm  : int16 [n_frames][1152][2] interleaved (stereo only: mono encode raises in the reference)
 * mdct : int32 [n_frames][2 ch][2 gr][576]  (the reference's __mdct_freq layout) */
int mp3s_encode_transform_dev(mp3s_ctx *ctx, const int16_t *d_pcm, const mp3s_frame_hdr *d_hdr, int n_frames,
                              int32_t *d_mdct);
int mp3s_encode_transform(mp3s_ctx *ctx, const int16_t *pcm, const mp3s_frame_hdr *hdr, int n_frames,
                          int32_t *mdct);

/* ---------------------------------------------------------------- (iv) rate-loop batch
 * replaces: __iteration_loop body per granule*channel: xrsq/xrabs/xrmax, __calc_scfsi energies,
 *           __bin_search_step_size, quantize, calc_run_len, count1_bit_count, __subdivide,
 *           __big_v_tab_select/__new_choose_table (+ IDX_TO_TRANSFORM_HUF hide swap), count_bit,
 *           big_v_bit_count, __inner_loop -- reference encoder/MP3_Encoder.py:171-318, 373-449, 760-1095,
 *           1147-1264.
 * Units are indexed u = (frame*2 + ch)*2 + gr, i.e. the reference's processing order (ch outer, gr inner).
 * mdct      : int32 [n_frames][2][2][576]
 * frames    : mp3s_rate_frame [n_frames]
 * hide_bits : n_hide bytes of 0/1 (NULL/0: not hiding); a unit reads hide_bits[i] only for i < min(n_hide,
 *             frames[frame].hide_end), so the messages of several streams can sit back to back in one array
 * cursor_in : int32 [n_units] hide-string index at the start of each unit (ignored when not hiding)
 * state_in  : int32 [n_units][4] address1, address2, address3, quantizerStepSize inherited from the same
 *             (gr,ch) of the previous frame (NULL = zeros)
 * unit_list : optional int32 [n_list] subset of units to (re)compute; NULL = all n_frames*4
 * ix        : int16 [n_frames][2][2][576] signed quantised spectrum (sign from mdct, :1272-1276)
 * out       : mp3s_gr_out [n_units]
 * en        : int32 [n_units][22] __calc_scfsi energies: 21 scalefactor bands + the granule total
 * The serial hide cursor / address chain is resolved by the caller (mp3s_encode_pcm does it):
 * run, prefix-sum n_tables, re-run the units whose assumed inputs were wrong, until none. */
int mp3s_rate_loop_dev(mp3s_ctx *ctx, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                       const uint8_t *d_hide_bits, int n_hide, const int32_t *d_cursor_in, const int32_t *d_state_in,
                       const int32_t *d_unit_list, int n_list, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en);

/* ---------------------------------------------------------------- (iv-b) the serial chains of the rate loop, on the device
 * replaces: the two values the reference carries from unit to unit -- __hide_str_offset (reference
 *           encoder/MP3_Encoder.py:808-809, read at :1154-1168) and the address1/2/3 + quantizerStepSize a granule
 *           without big values inherits from the same (gr, ch) of the frame before (:1004-1006, :788-803).
 * mp3s_rate_loop_dev runs every unit on ASSUMED inputs (cursor_in, state_in).  This call scans the batch (segments =
 * streams), compares every unit's assumptions with the true chain, gives silent units the values they inherit (the
 * bit packer writes them into the side info) and leaves one verdict: verdict[0] = number of units that ran on wrong
 * assumptions that mattered (0: d_gr is final, mp3s_pack_frames_dev may run), verdict[1] = 1 if some quantizer step
 * left its table.  The host then reads 8 bytes instead of 72 per unit. */
typedef struct {
    int32_t first_frame, n_frames;   /* the stream's frames in the batch */
    int32_t hide_base;               /* its message is hide[hide_base .. hide_end) of the batch's message array */
    int32_t hide_begin;              /* cursor at its first unit: hide_base + the bits taken by the frames in front of a block */
    int32_t hide_end;
    int32_t reserved;
    int32_t chain_in[4][4];          /* per ch*2+gr: address1..3, quantizerStepSize the frames in front left (zeros: stream start) */
} mp3s_chain_seg; /* 88 bytes */
typedef struct {
    int64_t cursor;                  /* behind the stream's last unit, as an index into the batch's message array */
    int32_t chain[4][4];             /* the four chains as the stream leaves them */
    int32_t carry_used;              /* some unit looked at chain_in / the message was live at the first unit */
    int32_t reserved;
} mp3s_chain_seg_out; /* 80 bytes */
/* d_state_in NULL = zeros (as for mp3s_rate_loop_dev); d_verdict: 2 words; d_seg_out: [n_segs]; both written by the call */
int mp3s_chain_resolve_dev(mp3s_ctx *ctx, mp3s_gr_out *d_gr, const mp3s_rate_frame *d_frames, int n_frames,
                           const mp3s_chain_seg *d_segs, int n_segs, const int32_t *d_cursor_in, const int32_t *d_state_in,
                           int32_t *d_verdict, mp3s_chain_seg_out *d_seg_out);

/* The same check, and what it finds wrong put right on the device where that takes one more run: the units that read
 * inherited addresses other than the chain really holds (the first pass gives every unit zeros: SURVEY E7 -- the first
 * quiet granules behind a silence), or ran on another cursor, are listed on the device, run again on the cursor and
 * addresses the check found (results in place, d_cursor updated for them) and everything is checked once more.  The
 * verdict is the second check's: 0 = final.  What it still finds (a re-run changed what later units inherit, more than
 * 1 024 units listed) is left to the host as before.  Operands as for mp3s_rate_loop_dev / mp3s_chain_resolve_dev; every
 * unit must have run without d_state_in (the records carry what they were given). */
int mp3s_chain_redo_dev(mp3s_ctx *ctx, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                        const uint8_t *d_hide_bits, int n_hide, int32_t *d_cursor, const mp3s_chain_seg *d_segs, int n_segs,
                        int16_t *d_ix, mp3s_gr_out *d_gr, int32_t *d_en, int32_t *d_verdict, mp3s_chain_seg_out *d_seg_out);

/* ---------------------------------------------------------------- (iv-c) the message cursor, decided on the device
 * replaces: the same __hide_str_offset chain (reference encoder/MP3_Encoder.py:808-809, :1154-1168), without guessing:
 *           a unit sees the message only through the <= 3 bits at its cursor (__new_choose_table reads hide_str[offset] once per non-zero table, :1257-1263), so the first
 *           `reach` units of a hiding stream are run once per possibility -- the 8 three-bit patterns, "two bits left",
 *           "one bit left" -- as extra entries of the SAME rate-loop launch, every unit's own run (cursor behind every
 *           message) being the eleventh, and a small kernel walks the chain and copies the entry each unit really sees
 *           into its place.  mp3s_chain_resolve_dev still checks the result; a stream whose message reaches further than
 *           the plan covered fails that check and is resolved by the host as before.  The walk goes in rounds of 1 024
 *           units with the state carried from round to round, so a plan may cover any length of message (r02f; until
 *           then 2 048 units, about 700 message bytes).
 * The message array must start with the 32 pattern bytes of mp3s_select_patterns() (so hide_base >= 32), and d_cursor
 * must hold MP3S_NO_CURSOR for every unit of a planned stream before the call; the call overwrites the cursors of the
 * units it replaced with what they really saw (a later call with the same plan may find those there: they are exact). */
#define MP3S_SELECT_VARIANTS 10
/* The two "bits left" possibilities can only meet the units around the message's end: a unit never takes more than three
 * tables, so unit j's cursor is at most 3j bits behind the start, and "two bits left" needs 3j >= bits_left - 2.  A stream's
 * entries are therefore 8 rows of `reach` units (the patterns) and 2 rows of the units from MP3S_SELECT_TAIL_FIRST on:
 *   entry (v < 8, j) = first_entry + v * reach + j
 *   entry (v >= 8, j >= t) = first_entry + 8 * reach + (v - 8) * (reach - t) + (j - t),  t = MP3S_SELECT_TAIL_FIRST(bits left, reach)
 * with bits left = hide_end - hide_begin of the stream's mp3s_chain_seg. */
#define MP3S_SELECT_TAIL_FIRST(bits_left, reach) ((bits_left) < 2 ? 0 : (((bits_left) - 2) / 3 < (reach) ? ((bits_left) - 2) / 3 : (reach)))
#define MP3S_SELECT_ENTRIES(bits_left, reach) (8 * (reach) + 2 * ((reach) - MP3S_SELECT_TAIL_FIRST(bits_left, reach)))
#define MP3S_SELECT_MAX_REACH (1 << 18)
#define MP3S_NO_CURSOR 0x3fffffff     /* "behind every message": such a unit hides nothing */
typedef struct {
    int32_t first_entry;             /* the stream's first entry (layout: MP3S_SELECT_TAIL_FIRST above) */
    int32_t reach;                   /* its first `reach` units are planned (0: the stream is left to cursor_in as it is) */
} mp3s_select_span; /* 8 bytes */
void mp3s_select_patterns(uint8_t out[32]);
/* host only: spans[n_segs] and the entries (unit, cursor; capacity `cap` each) for the streams of segs.  A stream that does
 * not hide, or whose message could reach more than MP3S_SELECT_MAX_REACH units, or that does not fit into cap any more
 * gets reach 0.  Returns the number of entries. */
int mp3s_select_plan(const mp3s_chain_seg *segs, int n_segs, mp3s_select_span *spans, int32_t *ent_unit, int32_t *ent_cursor, int cap);
/* mp3s_rate_loop_dev over all units of the batch (no inherited state) with the entries behind them in the same launch, then
 * the selection.  d_ixv / d_env: int16 [n_entries][576], int32 [n_entries][22]; d_outv: MP3S_VARIANT_OUT_BYTES(n_entries) --
 * mp3s_gr_out [n_entries] followed by the entries' table counts, one byte each. */
#define MP3S_VARIANT_OUT_BYTES(n) ((size_t)(n) * sizeof(mp3s_gr_out) + (((size_t)(n) + 15) & ~(size_t)15))
int mp3s_rate_select_dev(mp3s_ctx *ctx, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                         const uint8_t *d_hide_bits, int n_hide, int32_t *d_cursor, const mp3s_chain_seg *d_segs,
                         const mp3s_select_span *d_spans, int n_segs, int max_reach, const int32_t *d_ent_unit,
                         const int32_t *d_ent_cursor, int n_entries, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en,
                         int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env);
/* The two halves of mp3s_rate_select_dev on their own, for callers that put the selection (two small launches) on another
 * context's stream than the rate loop -- e.g. with the chain check and the bit packing under the decode transforms of the
 * next batch (bench.py).  The caller orders the streams (mp3s_ctx_wait). */
int mp3s_rate_variants_dev(mp3s_ctx *ctx, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                           const uint8_t *d_hide_bits, int n_hide, const int32_t *d_cursor, const int32_t *d_ent_unit,
                           const int32_t *d_ent_cursor, int n_entries, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en,
                           int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env);
int mp3s_select_dev(mp3s_ctx *ctx, const uint8_t *d_hide_bits, int32_t *d_cursor, const mp3s_chain_seg *d_segs,
                    const mp3s_select_span *d_spans, int n_segs, int max_reach, const int32_t *d_ent_unit,
                    const int32_t *d_ent_cursor, int n_entries, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en,
                    int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env);

/* ---------------------------------------------------------------- (vi) bit-level stages on the device
 * The serial bit parsing / packing of the reference is serial per granule only: granule boundaries are known from
 * the side info (decode) or from the rate loop (encode), so granules are decoded / packed in parallel.  The host
 * keeps the byte-level framing scan (sync, header, 17/32-byte side info, reservoir gather: mp3s_scan_stream). */

/* side info of one granule*channel as parsed from the stream (reference decoder/FrameSideInformation.py:62-137) */
typedef struct {
    uint16_t part2_3_length, big_values;
    uint8_t global_gain, scalefac_compress, window_switching, block_type, mixed_block_flag;
    uint8_t table_select[3];
    uint8_t region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    uint8_t sub_block_gain[3];
} mp3s_unit_side; /* 20 bytes */

/* one frame for the Huffman kernel: where its main data sits in the blob + the parsed side info
 * flags & MP3S_FS_HOST_DECODED: the kernel skips the frame (the caller places host-decoded samples there afterwards: the
 * last frame of a stream written by the reference's encoder can lack up to 3 bytes of its main data) */
#define MP3S_FS_HOST_DECODED 1
typedef struct {
    uint32_t md_off, md_len;   /* byte offset (multiple of 4) and length of the frame's main data in the blob */
    uint8_t nch, sr_idx, ms_stereo, flags;
    uint8_t scfsi[2][4];
    mp3s_unit_side unit[2][2]; /* [gr][ch] */
    int32_t reserved;          /* index (in the batch) of the first frame of this frame's stream; the scan writes 0.  Negative when the
                                * side array holds records of the stream in FRONT of the batch (mp3s_stream_ref.side_back) */
} mp3s_frame_side; /* 104 bytes */

/* replaces: __unpack_scale_fac + __unpack_samples for every granule*channel of the batch -- reference
 * decoder/Frame.py:365-559 (linear code-book search there, two-level tables here; same prefix codes, same quirks D1/D2).
 * blob: main data of all frames (reservoir already gathered), each frame 4-byte aligned and followed by >= 8 zero
 * bytes; is / si as consumed by mp3s_decode_transform_dev; status: int32, OR of MP3S_HS_* on malformed input.
 * Streams whose scalefactors are inherited across frames (mixed blocks, scfsi behind a short granule 0: SURVEY D10;
 * mp3s_scan_stream reports them with gpu_ok = 0): the kernel walks back through the stream's side records to the granule
 * that wrote the entry last and reads it from that granule's bits, so d_side / d_blob must hold such a stream from its first
 * frame on and frame_side.reserved must name that frame's index relative to d_side[0] (0 for a single stream; negative for a
 * chunk whose predecessors' records lie in front of d_side in one file-wide array: the one-file calls, round 4).  Blocks of
 * such a stream that a rank decodes on its own are parsed on the host. */
#define MP3S_HS_BAD_REGION 1
#define MP3S_HS_BIG_VALUES 2
#define MP3S_HS_HINT 4 /* some part2_3_length exceeds max_part2_3_length: call again with a larger bound (or 0) */
#define MP3S_HS_OVERRUN 8 /* the big values of some granule run past its part2_3_length into the bits that follow: the
                           * reference decodes on (one bit cursor per frame); use mp3s_parse_stream for this stream */
/* max_part2_3_length: an upper bound on part2_3_length over the batch (the scan knows it: mp3s_scanned.max_part2_3_length),
 * or 0 for the format's limit of 4095.  It sizes the per-thread staging of the bit stream in LDS and with it the number of
 * resident wavefronts: 8 per CU for granules up to 900 bits, 2 at the limit. */
int mp3s_huffman_decode_dev(mp3s_ctx *ctx, const uint8_t *d_blob, const mp3s_frame_side *d_side, int n_frames, int nch,
                            int max_part2_3_length, int16_t *d_is, mp3s_granule_si *d_si, int32_t *d_status);

/* ---- the byte-level scan split in two: a walk from header to header on the host, everything else on the device
 * replaces: the same lines as mp3s_scan_stream -- reference decoder/MP3_Parser.py:68-80 and Frame.py:288-316 (the walk: frame
 *           sizes are a recurrence on the headers), decoder/FrameSideInformation.py:39-137 and Frame.py:318-363 (side info
 *           taken apart field by field, main data collected from where the bit reservoir left it: on the device, one
 *           wavefront per frame, from the file image as it is).
 * The walk takes REGULAR streams -- MPEG-1 Layer III frames whose reservoir pointers stay inside the file -- and reports
 * anything else (regular = 0: false syncs parsed as other layers, free-format headers that inherit fields, pointers in front
 * of the file): such a stream goes through mp3s_scan_stream, which follows the reference through every oddity. */
typedef struct {
    uint32_t file_off;         /* offset of the frame header in the byte image the device holds */
    uint32_t md_off;           /* where the frame's main data goes in the blob (multiple of 4) */
    uint16_t md_len;           /* ... and how long it is (reservoir bytes included) */
    uint16_t frame_size;       /* bytes the frame loop steps over (Frame.py:311-316) */
    uint16_t stream;           /* index into the batch's mp3s_stream_ref array */
    uint16_t flags;            /* MP3S_FS_HOST_DECODED: the Huffman kernel leaves the frame alone */
} mp3s_frame_ref; /* 16 bytes */
typedef struct {
    uint32_t base, end;        /* the stream's file image inside the batch's byte image: [base, end); bytes behind `end` read
                                * as zero (reference decoder/util.py:41-43) */
    uint32_t first_frame;      /* its first frame in the batch (mp3s_frame_side.reserved, mp3s_frame_hdr.stream_first) */
    uint32_t n_frames;
    uint16_t prev_size[9];     /* what Frame.__prev_frame_size holds in front of first_frame's gather (SURVEY D11 at a stream's start) */
    uint16_t side_back[2];     /* lo, hi: frames of this stream whose side records and main data lie in FRONT of d_side[first_frame] /
                                * below d_blob's offsets of this batch -- a chunk of a file whose earlier chunks left theirs in one
                                * file-wide array (round 4).  mp3s_frame_side.reserved becomes first_frame - side_back, and the
                                * Huffman kernel's walk for scalefactors inherited across frames goes back that far */
    uint16_t reserved;
} mp3s_stream_ref; /* 40 bytes */
typedef struct {
    int32_t regular;              /* 0: use mp3s_scan_stream for this stream (nothing else below is meaningful) */
    int32_t n_frames, nch, sampling_rate, bit_rate, dup_last_frame;
    int32_t max_part2_3_length;   /* over all granules: the bound mp3s_huffman_decode_dev wants */
    int32_t any_silent;           /* some granule has no code book in use (no big values, or book 0 in all its regions) */
    size_t blob_len;              /* bytes of blob the frames take */
    const mp3s_frame_ref *refs;   /* [n_frames]: file_off counted from the start of the file, md_off from 0, stream 0 */
    mp3s_stream_ref stream;       /* base 0, end = len, first_frame 0 */
    const uint8_t *tables;        /* [n_frames][4]: code books in use per granule in the encoder's unit order (frame, ch, gr); stereo only */
} mp3s_walked;
int mp3s_walk_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_walked *out);
/* d_image[0] / d_blob[0] are byte image_base / md_base of what the references count in (a chunk of a long file brings its own
 * piece of both); d_tsel optional: per frame the twelve table indices, five bits each in the order the stego bits walk them
 * (channel, granule, region), the four window-switching flags above them -- mp3s_stego_bits turns them into the bits;
 * d_status: one zeroed word, MP3S_PS_* are OR-ed in.  Records as mp3s_scan_stream writes them, except table_select[2] of
 * window-switching granules and sub_block_gain of the others: a granule keeps those from the frame before,
 * nothing on the device reads them, they are zero here. */
#define MP3S_PS_INHERITS 1   /* scalefactors are inherited across frames somewhere (mp3s_scanned.gpu_ok == 0) */
#define MP3S_PS_MISMATCH 2   /* a frame's main data came out another length than its reference says */
int mp3s_parse_frames_dev(mp3s_ctx *ctx, const uint8_t *d_image, uint32_t image_base, const mp3s_frame_ref *d_refs,
                          const mp3s_stream_ref *d_streams, int n_frames, uint32_t md_base, mp3s_frame_side *d_side,
                          mp3s_frame_hdr *d_hdr, uint8_t *d_blob, uint64_t *d_tsel, int32_t *d_status);
/* replaces: __get_frame_huffman_tables + bit_from_huffman_tables -- reference decoder/Frame.py:676-685, decoder/util.py:67-81.
 * carry[4]: the third table index each (channel, granule) was left with by the frames in front (zeros at a stream's start), updated */
int mp3s_stego_bits(const uint64_t *tsel, int64_t n_frames, int nch, uint8_t carry[4], mp3s_buf **owner, const uint8_t **bits, size_t *n_bits);

/* replaces: __format_bitstream for every frame of the batch -- reference encoder/MP3_Encoder.py:1097-1145 (stuffing),
 * 1266-1547; one workgroup per frame, one wavefront per granule*channel, code lengths prefix-summed across lanes.
 * gr must be final (serial chain resolved, silent units carrying their inherited quantizer_step); en as written by
 * the rate loop (scfsi is decided here, :861-892); frame_off[f] = byte offset of frame f, padding[f] its padding bit.
 * Output bytes [frame_off[n_frames]]; the caller truncates the stream to a multiple of 4 bytes (E14). */
int mp3s_pack_frames_dev(mp3s_ctx *ctx, const int16_t *d_ix, const mp3s_gr_out *d_gr, const int32_t *d_en, int n_frames,
                         int samplerate, int bitrate_kbps, const uint32_t *d_frame_off, const uint8_t *d_padding,
                         uint8_t *d_mp3, int32_t *d_scfsi, int32_t *d_status);

/* ---------------------------------------------------------------- host stages (no GPU needed) */

/* replaces: the byte-level part of MP3Parser.parse_file: sync/header/side-info parse, frame sizes, reservoir gather,
 * stego bits -- reference decoder/MP3_Parser.py:25-85, Frame.py:244-263, 288-363, 676-685, FrameHeader.py,
 * FrameSideInformation.py.  No scalefactor / Huffman decoding (that is mp3s_huffman_decode_dev or mp3s_parse_stream). */
typedef struct {
    int32_t n_frames, nch, sampling_rate, bit_rate, n_bits, dup_last_frame;
    int32_t gpu_ok;               /* 0: scalefactors are inherited across frames somewhere -> use mp3s_parse_stream */
    int32_t max_part2_3_length;   /* over all granules: the bound mp3s_huffman_decode_dev wants */
    const mp3s_frame_side *side;  /* [n_frames] */
    const mp3s_frame_hdr *hdr;    /* [n_frames] */
    const uint8_t *blob;          /* main data of all frames */
    size_t blob_len;
    const uint8_t *bits;          /* stego bits 0/1 */
    const int32_t *frame_size;    /* [n_frames] */
} mp3s_scanned;
int mp3s_scan_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_scanned *out);

void mp3s_buf_free(mp3s_buf *b);

/* replaces: MP3Parser.parse_file front end: header/side-info parse, reservoir reassembly, scalefactor
 * and Huffman decode, stego bit extraction -- reference decoder/MP3_Parser.py:25-85, Frame.py:244-263,
 * 288-559, 676-685, FrameHeader.py, FrameSideInformation.py, util.py:22-81, ID3 skip (ID3_Parser.py:95-131). */
typedef struct {
    int32_t n_frames, nch, sampling_rate, bit_rate; /* rate/bitrate of the LAST header */
    int32_t n_bits;          /* stego bits */
    int32_t dup_last_frame;  /* reference appends the last PCM frame once more after a bad header (D12) */
    const int16_t *is;       /* [n_frames][2][2][576] */
    const mp3s_granule_si *si;
    const mp3s_frame_hdr *hdr;
    const uint8_t *bits;     /* 0/1 */
    const int32_t *table_select; /* [n_frames][2 gr][2 ch][3] as parsed (for tests) */
    const int32_t *frame_size;   /* [n_frames] */
} mp3s_parsed;
int mp3s_parse_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_parsed *out);

/* replaces: Encoder back end: padding/slot-lag replay, __resv_frame_end, __format_bitstream and the
 * 32-bit cache writer -- reference encoder/MP3_Encoder.py:623-636, 1097-1145, 1266-1552.
 * ix/gr/scfsi as produced by the rate loop for n_frames stereo frames. */
int mp3s_format_stream(int samplerate, int bitrate_kbps, int n_frames, const int16_t *ix, const mp3s_gr_out *gr,
                       const int32_t *scfsi /*[n_frames][2][4]*/, mp3s_buf **owner, const uint8_t **mp3, size_t *mp3_len);
/* per-frame padding / max_bits replay (reference MP3_Encoder.py:630-636, 894-912) */
int mp3s_rate_frames(int samplerate, int bitrate_kbps, int nch, int n_frames, mp3s_rate_frame *out, int32_t *padding);

/* ---------------------------------------------------------------- (v) whole-stream conveniences */
/* replaces: Decoder.decode up to (not including) the WAV write -- reference decoder/decoder.py:59-84 */
typedef struct {
    int32_t n_frames, nch, sampling_rate, bit_rate, n_bits;
    int64_t n_rows;      /* PCM rows = 1152 * (n_frames + dup_last_frame) */
    const void *pcm;     /* [n_rows][nch] in out_format */
    const uint8_t *bits; /* stego bits 0/1 */
} mp3s_decoded;
int mp3s_decode_stream(mp3s_ctx *ctx, const uint8_t *file, size_t len, int out_format, mp3s_buf **owner,
                       mp3s_decoded *out);
/* many streams in one call: the frames of all files form one batch (stream_first marks the starts), so
 * a corpus of short files costs one Huffman launch and one transform launch per channel count instead of one per file.
 * out[i] describes file i; every pointer lives in *owner.  status[i] = MP3S_OK or the code file i alone would have
 * failed with (its out[i] is zeroed, the other files are unaffected); with status == NULL the first such file fails the
 * whole call (its index is in mp3s_last_error()) -- the same rule as mp3s_hide_messages. */
int mp3s_decode_streams(mp3s_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n_files, int out_format,
                        mp3s_buf **owner, mp3s_decoded *out, int32_t *status);

/* Frames [first_frame, first_frame + n_frames) of the stream (clipped to its end) -- the decode half of sharding one
 * stream over several GPUs.  The whole file is scanned on the host; only the block's main data plus one frame
 * in front of it goes to the device (IMDCT overlap and synthesis fifo reach back less than a frame: reference
 * decoder/Frame.py:151-153, 81-92).  Concatenating the blocks' PCM gives mp3s_decode_stream's.  n_frames / n_rows
 * describe the block; bits / bit_rate / sampling_rate the whole stream. */
int mp3s_decode_block(mp3s_ctx *ctx, const uint8_t *file, size_t len, int64_t first_frame, int64_t n_frames, int out_format,
                      mp3s_buf **owner, mp3s_decoded *out);

/* Index of a stream: ONE walk over its frames (headers, side info, reservoir pointers; no main data is copied) that leaves
 * a resume point every 256 frames.  With it a block of the stream is scanned on its own -- the ranks of a sharded stream
 * and the chunks of a streamed file pay for their own frames (plus at most 255 skipped ones), not for the whole file again
 * (reference: none -- it reads the whole file into memory, decoder/decoder.py:26-27, and loops over it once,
 * decoder/MP3_Parser.py:68-80).  The index is plain host data, bound to the file bytes it was made from. */
typedef struct mp3s_index mp3s_index;
typedef struct {
    int64_t n_frames;
    int32_t nch, sampling_rate, bit_rate;   /* of the LAST frame header, as mp3s_scan_stream reports them */
    int32_t dup_last_frame;
    int32_t gpu_ok;                         /* 0: a stream the host parser has to take (mp3s_scanned.gpu_ok): block calls then scan the whole file */
    int32_t reserved;
} mp3s_index_info;
int mp3s_index_stream(const uint8_t *file, size_t len, mp3s_index **index, mp3s_index_info *info);
void mp3s_index_free(mp3s_index *index);
/* mp3s_scan_stream for frames [first_frame, first_frame + n_frames) only (clipped to the stream): side records, main-data
 * blob, headers, frame sizes and the stego bits OF THESE FRAMES, identical to the corresponding part of the full scan */
int mp3s_scan_range(const uint8_t *file, size_t len, const mp3s_index *index, int64_t first_frame, int64_t n_frames, mp3s_buf **owner,
                    mp3s_scanned *out);
/* mp3s_decode_block with the block scanned on its own; out->bits / n_bits then describe the block's frames (the halo frame in
 * front included), not the whole stream.  index == NULL: exactly mp3s_decode_block. */
int mp3s_decode_block_indexed(mp3s_ctx *ctx, const uint8_t *file, size_t len, const mp3s_index *index, int64_t first_frame,
                              int64_t n_frames, int out_format, mp3s_buf **owner, mp3s_decoded *out);

/* replaces: Encoder.encode -- reference encoder/encoder.py:33-58, MP3_Encoder.py:596-618 */
typedef struct {
    int32_t n_frames;
    int32_t too_long;      /* hide_str_offset < len(hide_str) - 1 (encoder.py:50) */
    int64_t hide_offset;
    const uint8_t *mp3;
    size_t mp3_len;
    const mp3s_gr_out *gr; /* [n_frames*4] unit order (frame, ch, gr) */
    const int32_t *scfsi;  /* [n_frames][2][4] */
    int32_t rate_passes;   /* launches of the rate loop needed to resolve the serial chain */
} mp3s_encoded;
int mp3s_encode_pcm(mp3s_ctx *ctx, const int16_t *pcm, int64_t n_samples_per_ch, int nch, int samplerate,
                    int bitrate_kbps, const uint8_t *hide_bits, int n_hide, mp3s_buf **owner, mp3s_encoded *out);

/* One contiguous block of a longer stream, for a stream sharded over several GPUs.  What crosses a block
 * boundary in the reference encoder: < 1 frame of filter-bank / MDCT history (MP3_Encoder.py:356,685,747: `lead_frames`
 * = 1 frame of PCM in front of the block is transformed and dropped), the padding recurrence (:630-636: restarted from
 * frame 0 by `first_frame`), and two serial chains, passed explicitly: */
typedef struct {
    int64_t cursor;        /* message bits consumed by the frames in front of the block (MP3_Encoder.py:808-809) */
    int32_t chain[4][4];   /* per ch*2+gr: address1, address2, address3, quantizerStepSize as the frame in front left
                            * them (they are only re-written by granules that carry data: SURVEY E7) */
} mp3s_carry;
/* pcm        : int16 [(lead_frames + n) * 1152][2], n = the block's own frames; stereo only
 * first_frame: index of the block's first own frame in the stream;  last_block: the stream ends with this block
 * hide_bits  : the WHOLE message (cursors are absolute);  carry_in NULL = the block starts the stream
 * carry_out  : what the next block needs;  *carry_used = the block's bytes depend on carry_in (a caller that ran the
 *              block on a guessed carry_in re-runs it only if this is set and the guess was wrong)
 * out        : mp3 = the block's frames (concatenating the blocks gives the stream mp3s_encode_pcm produces),
 *              hide_offset counts from the start of the stream; too_long is meaningful for the last block */
int mp3s_encode_block(mp3s_ctx *ctx, const int16_t *pcm, int64_t n_samples_per_ch, int lead_frames, int64_t first_frame,
                      int last_block, int samplerate, int bitrate_kbps, const uint8_t *hide_bits, int n_hide,
                      const mp3s_carry *carry_in, mp3s_carry *carry_out, int32_t *carry_used, mp3s_buf **owner,
                      mp3s_encoded *out);

/* ---------------------------------------------------------------- (vi) files and messages */
/* Whole files as byte strings in, byte strings out: WAV and MP3 containers, message framing and the facade's three
 * operations inside the library, so that a binding in any language needs no code of its own for them.  Nothing here
 * touches the disk.  MP3S_E_EXIT = the reference would sys.exit(text); MP3S_E_UNSUPPORTED / MP3S_E_MALFORMED = it would
 * raise (IndexError / struct.error ...). */

/* replaces: WavReader.__read_header + check_bitrate_index -- reference encoder/WAV_Reader.py:30-111 */
typedef struct {
    int32_t channels, samplerate, bits_per_sample, bitrate;
    int64_t num_of_samples; /* per channel, from the data chunk size (float arithmetic as in the reference) */
    int64_t data_offset;    /* byte offset of the first sample */
    int64_t n_values;       /* int16 values the reference's np.fromfile call yields (up to 2x the declared count) */
} mp3s_wav_info;
int mp3s_wav_parse(const uint8_t *file, size_t len, int bitrate_kbps, mp3s_wav_info *out);
/* replaces: scipy.io.wavfile.write header as used by MP3_Parser.write_to_wav -- reference decoder/MP3_Parser.py:86-93 */
int mp3s_wav_header(int64_t n_rows, int nch, int rate, uint8_t out44[44]);
/* replaces: str_to_binary_str(str(len(m)) + "#" + m) -- reference steganography.py:10-24, 42-50.  bits are 0/1 bytes. */
int mp3s_message_frame(const uint8_t *utf8, size_t n, mp3s_buf **owner, const uint8_t **bits, size_t *n_bits);
/* replaces: the reveal parse -- reference decoder/decoder.py:90-108.  text is what the reference writes to the .txt. */
int mp3s_message_reveal(const uint8_t *bits, size_t n_bits, mp3s_buf **owner, const uint8_t **text, size_t *n_text);

typedef struct {
    const uint8_t *data;  /* the produced file: WAV, MP3 or revealed text */
    size_t len;
    int32_t kbps;         /* bitrate of the last frame header / 1000 (decoder.py:110); the bitrate used when encoding */
    int32_t sampling_rate, channels, n_frames;
    int32_t too_long;     /* encode/hide: the message did not fit and was trimmed (encoder.py:50) */
    int32_t n_bits;       /* decode: stego bits found in the stream */
    int64_t hide_offset;  /* encode/hide: message bits consumed */
    const uint8_t *bits;  /* decode: the stego bits, 0/1 */
} mp3s_file;
/* replaces: Decoder(...).decode() -- reference decoder/decoder.py:59-84: MP3 bytes -> WAV bytes (int16) */
int mp3s_decode_file(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out);
/* the same with the WAV written to an open file (decoder/decoder.py:80-84 ends in scipy.io.wavfile.write(output_file_path, ...)): bytes
 * [0, out->len) of `fd` (pwrite), cut to that length; the PCM of a file that goes through the stages as chunks is written chunk by chunk
 * while the later chunks are on the device (same rule as mp3s_hide_message_fd for a file that held something).  out->data is NULL, out->bits
 * lives in *owner. */
int mp3s_decode_file_fd(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, int fd, mp3s_buf **owner, mp3s_file *out);
/* replaces: Encoder(...).encode() -- reference encoder/encoder.py:21-58: WAV bytes -> MP3 bytes, hide_bits optional */
int mp3s_encode_file(mp3s_ctx *ctx, const uint8_t *wav, size_t len, int bitrate_kbps, const uint8_t *hide_bits,
                     int n_hide, mp3s_buf **owner, mp3s_file *out);
/* replaces: Steganography.hide_message / clear_file -- reference steganography.py:133-182: decode, then re-encode at
 * the stream's own bitrate with (hide) or without (clear) "<count>#<message>".  The int16 PCM never leaves HBM between
 * the two halves; the result is byte-identical to going through the temporary WAV file. */
int mp3s_hide_message(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, mp3s_buf **owner,
                      mp3s_file *out);
int mp3s_clear_file(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out);
/* The same two calls with the result written to an open file instead of handed back (the reference's hide_message / clear_file write
 * a file: steganography.py:137-162, 164-182 -- the last step of both is Encoder(...).encode() writing output_file_path,
 * encoder/encoder.py:53-57): bytes [0, out->len) of `fd` are the result (pwrite: the descriptor's position is not used or moved), the
 * file is cut to that length at the end; out->data is NULL.  A file that goes through the stages as chunks is written chunk by chunk, a
 * chunk's bytes while the chunks behind it are on the device -- if the file was EMPTY when the call began; a file that held something is
 * written at the end of a call that succeeded and is left as it was by one that did not (the reference has not touched its output when it
 * refuses a stream; whoever created an empty file for the call removes it after an error). */
int mp3s_hide_message_fd(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int fd, mp3s_file *out);
int mp3s_clear_file_fd(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, int fd, mp3s_file *out);
/* replaces: a loop of Steganography.hide_message / clear_file over a list of files.  All files with the
 * same sampling rate and bitrate go through the device as ONE batch (decode, transforms, rate loop, bit packing: the
 * streams' frames back to back, every serial chain restarting at a stream's first frame), so many short files cost
 * about what one long file of the same total length costs.  msgs[i] = UTF-8 message of file i, NULL = clear that file
 * (msgs itself NULL = clear all).  out[i] is byte-identical to what mp3s_hide_message / mp3s_clear_file give for file i
 * alone.  status[i] = MP3S_OK or the code file i alone would have failed with (its out[i] is zeroed); with status ==
 * NULL the first such code fails the whole call. */
int mp3s_hide_messages(mp3s_ctx *ctx, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                       const size_t *msg_lens, mp3s_buf **owner, mp3s_file *out, int32_t *status);
/* The share of one rank in hiding a message in (utf8 != NULL) or clearing (NULL) ONE stream that is spread over `world`
 * ranks: the stream's PCM frames (repeated last frame included) are cut into `world` contiguous blocks, sizes
 * differing by at most one; this call decodes block `rank` -- one frame of decoder state and one frame of PCM in front of
 * it -- and re-encodes it without the PCM leaving HBM: mp3s_decode_block + mp3s_encode_block in one, with one host scan.
 * carry_in NULL is only valid for rank 0; the other ranks pass the carry_out of the rank before them -- or a guess, and
 * call again with the real one if carry_used says the block depended on it (see mp3s_encode_block).  file.data = the
 * block's frames; concatenated over the ranks they are the file mp3s_hide_message / mp3s_clear_file produce.  A rank
 * beyond the last frame gets n_frames = 0 and no data. */
typedef struct {
    int64_t total_frames, first_frame, n_frames; /* of the stream / of this rank's block */
    int32_t is_last, carry_used;
    mp3s_carry carry_out;
    mp3s_file file;                              /* too_long / hide_offset count from the start of the stream */
} mp3s_block;
int mp3s_reencode_block(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int rank, int world,
                        const mp3s_carry *carry_in, mp3s_buf **owner, mp3s_block *out);
/* mp3s_reencode_block with the block scanned on its own (index == NULL: exactly mp3s_reencode_block) */
int mp3s_reencode_block_indexed(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, const mp3s_index *index, const uint8_t *utf8, size_t n_msg,
                                int rank, int world, const mp3s_carry *carry_in, mp3s_buf **owner, mp3s_block *out);
/* mp3s_hide_message (utf8 != NULL) / mp3s_clear_file (NULL) on a file of any length in chunks of at most chunk_frames frames:
 * one index walk, then chunk after chunk through the device, each scanned on its own and run on the real carry of the one in
 * front of it (streaming for files larger than the device / page-locked buffers should take at once).  The
 * device and staging memory in use is that of one chunk; the result is byte-identical to the one-call functions. */
int mp3s_hide_message_chunked(mp3s_ctx *ctx, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int64_t chunk_frames,
                              mp3s_buf **owner, mp3s_file *out);
/* replaces: Steganography.reveal_massage -- reference steganography.py:103-131: MP3 bytes -> message text.  Only the
 * byte-level scan runs (table_select lives in the side info), so no device work and no context are needed.
 * More lenient than the reference in one respect: reveal_massage decodes every frame before it looks at the bits, so a
 * stream whose main data makes that decode raise (big_values > 288, region counts past the band table, ragged channel
 * counts are caught here too, Huffman-level damage is not) raises there and yields a message here.  The Python facade is
 * unaffected (Steganography.reveal_massage goes through mp3s_decode_file). */
int mp3s_reveal_message(const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out);

/* ---------------------------------------------------------------- (vii) asynchronous host-fed pipeline
 * replaces: a loop of Steganography.hide_message / clear_file over many files or batches of files -- reference
 *           steganography.py:133-182, whose two serial frame loops (decoder/MP3_Parser.py:68-80, encoder/MP3_Encoder.py:
 *           607-609) become four overlapping stages: host scan on worker threads (straight into page-locked staging) ||
 *           hipMemcpyAsync up || kernels || hipMemcpyAsync down, `depth` jobs in flight, nothing waited for until a result
 *           is collected.
 * A job = the files of one mp3s_hide_messages call (same arguments, same results byte for byte: jobs the device cannot
 * take in one batch, or whose Huffman data is damaged, are redone by that very function; a job whose on-device verdict says
 * a guess failed keeps its device buffers and has its chains resolved at collect time).
 * While a pipe exists its context belongs to it: no other call may use the context until mp3s_pipe_destroy().
 * The submitted file and message buffers are borrowed until the job has been collected. */
typedef struct mp3s_pipe mp3s_pipe;
typedef struct {
    int64_t submitted, collected;
    int64_t fast, slow;               /* collected jobs that went through the overlapped stages / through mp3s_hide_messages */
    double scan_ms, issue_ms;         /* summed over jobs: host scan + input layout; queueing the job's device work */
    double last_device_span_ms;       /* first upload byte to last download byte of the job collected last (HIP events) */
    double scan_cpu_ms;               /* CPU time of the scan threads inside scan_ms (less than scan_ms: the threads were not running) */
    int64_t resolved;                 /* collected jobs whose cursor / address guess failed and whose chains the host resolved on the job's
                                       * own device buffers (scan, decode and transforms kept); fast + resolved + slow = collected */
    /* how the pipe's streams were chosen when it was made (the runtime maps streams onto a few hardware queues; a pipeline whose
     * stages share queues loses its overlap): `rehearsals` miniature jobs of about 1 ms each, at most 12 and 15 ms in all -- 13 x the one-stream miniature, within 60 ms, where that is more (0: a
     * pipe made earlier on this context's stream decided), judged against the same miniature on ONE stream */
    double rehearsal_ms;
    int64_t rehearsals;
    int64_t lanes;                    /* rotation of the copy / front-end candidates | (compute candidate + 1) << 8 | (tail candidate + 1) << 16 */
    int64_t queue_shared;             /* 1: no choice got the miniature through in 0.94 x the one-stream time (side by side: 0.86 - 0.89; on one queue: 1.1 - 1.4): stages share hardware queues */
} mp3s_pipe_stats;
/* depth: jobs in flight (= staging slots); max_job_bytes: MP3 bytes per job the staging is sized for (larger jobs still
 * work, through the synchronous path); scan_threads: host workers */
int mp3s_pipe_create(mp3s_ctx *ctx, int depth, size_t max_job_bytes, int scan_threads, mp3s_pipe **out);
void mp3s_pipe_destroy(mp3s_pipe *pipe);
/* arguments as for mp3s_hide_messages (msgs NULL = clear all, msgs[i] NULL = clear file i); MP3S_E_BUSY when `depth`
 * jobs are in flight (collect one first) */
int mp3s_pipe_submit(mp3s_pipe *pipe, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                     const size_t *msg_lens, int64_t *ticket);
/* a decode job: the files of one mp3s_decode_file loop, MP3 bytes -> WAV bytes (int16) + stego bits per file; results
 * through mp3s_pipe_collect like those of the other jobs (out[i].data = the WAV image, bits / n_bits set) */
int mp3s_pipe_submit_decode(mp3s_pipe *pipe, const uint8_t *const *mp3s, const size_t *lens, int n_files, int64_t *ticket);
/* waits for the OLDEST job in flight and hands out its results: out[i] / status[i] as mp3s_hide_messages fills them
 * (max_files = room in both arrays), *n_files = files of that job.  MP3S_E_BUSY: nothing in flight. */
int mp3s_pipe_collect(mp3s_pipe *pipe, int64_t *ticket, mp3s_buf **owner, mp3s_file *out, int32_t *status, int max_files,
                      int *n_files);
/* a block job: the share of rank `rank` of `world` in hiding a message in (utf8 != NULL) or clearing ONE stream -- the
 * arguments and the result of mp3s_reencode_block (carry_in NULL for rank 0; the carry is copied at submission) -- through
 * the overlapped stages beside the other jobs, so that the blocks a rank holds of streams that straddle its boundaries do
 * not interrupt the flow of the whole streams in front of and behind them (BASELINE configs[3]).  The result comes through
 * mp3s_pipe_collect_block when the job is the oldest in flight (mp3s_pipe_collect refuses a block job with MP3S_E_ARG). */
int mp3s_pipe_submit_block(mp3s_pipe *pipe, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int rank, int world,
                           const mp3s_carry *carry_in, int64_t *ticket);
int mp3s_pipe_collect_block(mp3s_pipe *pipe, int64_t *ticket, mp3s_buf **owner, mp3s_block *out);
/* 1: the oldest job in flight is a block job (collect it with mp3s_pipe_collect_block), 0: another job, MP3S_E_BUSY: none */
int mp3s_pipe_next_is_block(mp3s_pipe *pipe);
int mp3s_pipe_get_stats(mp3s_pipe *pipe, mp3s_pipe_stats *out);

#ifdef __cplusplus
}
#endif
#endif
