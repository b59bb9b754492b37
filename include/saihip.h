/*
 * saihip.h -- C ABI of libsaihip.so: sai's sliding-window U/Q statistics on MI355X (gfx950).
 *
 * The reference (xin-huang/sai 1.1.2) is pure Python and has no FFI; its plugin surface for this
 * path is three Python protocols (statistic classes, generators, preprocessors).  This header is
 * the one native boundary underneath the Python mirror of those protocols (package sai_amd): each
 * entry point states which reference function(s) it replaces, file:line relative to the
 * reference checkout.  INTEGRATION.md shows the ctypes binding a sai maintainer would add.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = error class (enum sai_status);
 *     sai_last_error() returns thread-local text for the last failure on the calling thread;
 *   - the caller owns every buffer; all data pointers are DEVICE pointers unless the name ends in
 *     _host; the library allocates nothing but the small sai_ctx and, inside it, the scratch of
 *     sai_single_window;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on
 *     it and no entry point synchronises unless documented;
 *   - a ctx is bound to one device and is not thread-safe; distinct ctxs are independent;
 *   - arguments are validated (sizes, NULLs, ranges); statistical parameters (w, x, y, quantile)
 *     are assumed already validated by the caller the way stat_utils.py:99-111 does.
 *
 * Data layout in HBM ("tiled SoA")
 *   A population's genotype block holds ALT-allele dosages as int8 (negative = missing call,
 *   sai/utils/utils.py:389-410 sums the ploidy axis so "./." arrives as -2).  Sites are grouped
 *   in tiles of SAI_TILE_SITES = 64 consecutive sites; inside a tile every individual owns one
 *   contiguous 64-byte row:
 *       byte(tile t, individual i, site-in-tile s) = tiles[(t * n_ind + i) * 64 + s]
 *   so one wavefront instruction (64 lanes x 16 B) reads 16 individuals x 64 sites = 1 KiB of
 *   contiguous memory and a tile is one contiguous n_ind * 64 byte run.  The last tile is padded
 *   with zero bytes.  sai_tiled_bytes() gives the allocation size.
 *
 * Flag planes and target frequencies (what the per-site decision hands to the windows stage)
 *   Per tile of 64 sites ONE row of 64-bit words, bit b of a word = site tile * 64 + b; for a call
 *   with n parameter sets:
 *       planes[tile * plane_stride + 0]           "any": sites whose tgt_freq is stored -- the OR of the
 *                                                 sets' conditions (SAI_FREQ_CANDIDATES), all ones (SAI_FREQ_DENSE)
 *       planes[tile * plane_stride + 1 + s]       condition of compute_matching_loci for set s (stat_utils.py:166)
 *       planes[tile * plane_stride + 1 + n + s]   site inverted for set s: its effective target frequency is
 *                                                 1 - tgt_freq (stat_utils.py:156-160).  Written, and read, only
 *                                                 when some set of the call has anc_allele_available == 0
 *                                                 (with ancestral alleles nothing is ever inverted).
 *   The target frequency of site (tile, b) with its "any" bit up lies at
 *       tgt_freq[tile * 64 + popcount(any & ((1 << b) - 1))]
 *   i.e. a tile's stored frequencies sit packed at the start of the tile's 64 slots (all ones: slot b).
 *   U's last test, tgt_freq > x (u_statistic.py:92), is taken by the windows stage from these
 *   frequencies: round 3's first layout also kept a "condition && tgt > x" plane and the inverted plane
 *   of every set, and each site's frequency in its own slot -- 7.5 + 8 partly written 64-byte lines per
 *   tile for C5's 20 sets where this layout writes 2.6 + 2.5, and in the middle of a read stream a
 *   written LINE is what costs.
 *   plane_stride = words per tile row, >= SAI_PLANES_PER_SET * n (room for 1 + 2 n words; a caller that
 *   evaluates more than SAI_MAX_SETS sets in several calls hands each call the row offset
 *   SAI_PLANES_PER_SET * first_set and must use SAI_FREQ_DENSE, whose slots do not depend on the sets).
 *   Bits of sites >= n_sites are 0 in the condition and inverted words.  A tile's row is written by one
 *   store instruction of the wavefront that evaluated the tile.
 */
#ifndef SAIHIP_H
#define SAIHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAI_ABI_VERSION 16
#define SAI_TILE_SITES 64
#define SAI_MAX_SRC 14  /* source populations of a parameter set (the reference loops over any number, stat_utils.py:114-119) */
#define SAI_FUSED_SRC 6 /* source populations a streaming pass takes per call (sai_site_counts, sai_site_pass, ...): more of
                           them go through sai_site_counts in groups + sai_site_flags, which takes 2 + SAI_MAX_SRC populations */
#define SAI_MAX_SETS 20 /* parameter sets per call: 1 + 2 * 20 plane words of a tile = one wave store (C5's 18 sets fit) */
#define SAI_FUSED_SETS SAI_MAX_SETS /* parameter sets the fused site pass carries */
#define SAI_PLANES_PER_SET 3 /* words of a tile row reserved per set (1 + 2 n <= 3 n are used) */

enum sai_status {
  SAI_OK = 0,
  SAI_ERR_ARG = -1,        /* bad argument (NULL, size, range) */
  SAI_ERR_HIP = -2,        /* a HIP runtime call failed */
  SAI_ERR_NO_DEVICE = -3,  /* no usable gfx950 device */
  SAI_ERR_UNSUPPORTED = -4 /* valid request outside the built limits */
};

/* comparison operators of the source-frequency conditions (stat_utils.py:133-139) */
enum sai_op { SAI_OP_EQ = 0, SAI_OP_LT = 1, SAI_OP_GT = 2, SAI_OP_LE = 3, SAI_OP_GE = 4 };
enum sai_freq_mode { SAI_FREQ_DENSE = 0, SAI_FREQ_CANDIDATES = 1 }; /* see sai_site_pass */

typedef struct sai_ctx sai_ctx;

/* One population block in the tiled SoA layout. */
typedef struct sai_pop {
  const int8_t* tiles; /* device pointer, sai_tiled_bytes(n_sites, n_ind) bytes */
  int32_t n_ind;       /* individuals (columns of the reference's [sites][individuals] matrix) */
  int32_t ploidy;      /* positive; calc_freq's `ploidy` argument (stat_utils.py:26) */
} sai_pop;

/* One parameter set = the arguments of compute_matching_loci (stat_utils.py:55-63) plus the
 * statistic-specific thresholds: x of UStatistic.compute (u_statistic.py:92) and quantile of
 * QStatistic.compute (q_statistic.py:100).  y[k]/op[k] pair with source population k;
 * one_minus_y[k] must be the caller's f64 evaluation of 1 - y[k] (stat_utils.py:150). */
typedef struct sai_params {
  double w;
  double x;
  double quantile;
  int32_t n_src;                /* must equal the number of source populations passed */
  int32_t anc_allele_available; /* 0: also match 1-y and invert (stat_utils.py:146-160) */
  int32_t op[SAI_MAX_SRC];      /* enum sai_op */
  double y[SAI_MAX_SRC];
  double one_minus_y[SAI_MAX_SRC];
} sai_params;

/* Per (parameter set, window) result: 24 bytes. */
typedef struct sai_window_record {
  int32_t n_sites; /* N(Variants): sites of the block inside [start, end] (feature_preprocessor.py:124) */
  int32_t u_count; /* U value (u_statistic.py:96) */
  int32_t n_cond;  /* sites passing compute_matching_loci's condition = size of the Q sample */
  int32_t n_cdd_q; /* sites with tgt_freq >= Q (q_statistic.py:101) */
  double q;        /* Q value, NaN when n_cond == 0 (q_statistic.py:96-100) */
} sai_window_record;

/* ---- library / context ------------------------------------------------------------------ */

int sai_abi_version(void);            /* SAI_ABI_VERSION the library was built with */
const char* sai_build_arch(void);     /* "gfx950" */
const char* sai_last_error(void);     /* thread-local, never NULL */
int sai_device_count(int* count_out); /* visible HIP devices */

/* Which physical GPU a HIP device index of this process is: its PCI bus id ("0000:05:00.0", at least 16
 * bytes of room) and, when uuid_hex_out is not NULL, its 16-byte UUID as 32 hex digits (33 bytes of room).
 * No reference counterpart: a job of one worker process per GPU (the MI355X form of mp_pool.py:45-73) reports
 * them per rank, so that its record shows N ranks on N distinct devices. */
int sai_device_identity(int device, char* bus_id_out, int32_t bus_id_capacity, char* uuid_hex_out,
                        int32_t uuid_capacity);

int sai_ctx_create(int device, sai_ctx** ctx_out);
int sai_ctx_destroy(sai_ctx* ctx);

/* ---- layout ----------------------------------------------------------------------------- */

/* Bytes of a tiled SoA block: ceil(n_sites / 64) * n_ind * 64.  Returns -1 on bad arguments. */
int64_t sai_tiled_bytes(int64_t n_sites, int32_t n_ind);

/* Re-tile a reference-order block.  `src` is the reference's per-population matrix
 * [site][individual] (window_generator.py:217-231 / utils.py:410) narrowed to int8 with
 * `row_stride` bytes between sites; `dst` receives the tiled SoA block (padding zeroed). */
int sai_tile_from_site_major(sai_ctx* ctx, const int8_t* src, int64_t n_sites, int32_t n_ind,
                             int64_t row_stride, int8_t* dst, void* stream);

/* ---- the hot path ----------------------------------------------------------------------- */

/* Kernel 1 (HBM-bound): per site and population, the integer part of calc_freq
 * (stat_utils.py:45-49): alt_sum = sum of non-negative dosages, n_called = individuals with a
 * non-negative dosage.  pops[0] = ref, pops[1] = tgt, pops[2..] = sources.
 * counts[(p * n_sites + site) * 2 + {0,1}] = {alt_sum, n_called}, uint32. */
int sai_site_counts(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                    uint32_t* counts, void* stream);

/* Words of a flag-plane buffer with rows of exactly SAI_PLANES_PER_SET * n_sets words:
 * ceil(n_sites / 64) * 3 * n_sets (-1 on bad arguments). */
int64_t sai_plane_words(int64_t n_sites, int32_t n_sets);

/* Kernels 1+2 fused (the fast path when there are at most SAI_FUSED_SETS parameter sets): one pass over the
 * genotypes that also evaluates sai_site_flags' per-site decision for each set at the end of every
 * tile, while the counts are still on chip.  `counts` may be NULL (then the 8 bytes per site and
 * population are neither written nor re-read); pops[p].ploidy is used.  Results are identical to
 * sai_site_counts followed by sai_site_flags.
 * freq_mode = SAI_FREQ_DENSE writes tgt_freq[site] for every site ("any" word all ones);
 * SAI_FREQ_CANDIDATES writes it only for sites at which some set's condition bit is up -- the only
 * entries sai_window_stats reads -- packed at the start of the tile's 64 slots in site order ("any" =
 * the OR of the conditions; see the top of this header) and leaves the rest of the buffer untouched
 * (dense 8-byte stores interleaved with the genotype stream cost about 10 % of the pass on MI355X,
 * a tile's candidates in slots of their own a fifth of that for sets as loose as C5's).
 * Streams: a launch of many multi-wavefront workgroups (sai_window_stats, sai_site_flags, any large kernel)
 * that waits in THIS pass's stream behind the running pass costs the pass ~6 % (profiles/history/r04_lone_pass.txt);
 * enqueue what follows on a second stream behind an event wait. */
int sai_site_pass(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                  uint32_t* counts, int32_t n_sets, const sai_params* sets_host, int32_t freq_mode,
                  double* tgt_freq, uint64_t* planes, int64_t plane_stride, void* stream);

/* Kernel 2: calc_freq's f64 division (stat_utils.py:51-52) and compute_matching_loci
 * (stat_utils.py:114-166) for every site and parameter set.  tgt_freq[site] is the UNinverted
 * target frequency (NaN when nothing is called), written for every site (SAI_FREQ_DENSE: "any" all
 * ones); the decisions go to the flag planes described at the top of this header (condition, site
 * inverted).  U's final test (u_statistic.py:92) is sai_window_stats'.
 * ploidy[p] pairs with population p of sai_site_counts; up to 2 + SAI_MAX_SRC populations (counts of more than
 * 2 + SAI_FUSED_SRC come from several sai_site_counts calls into one tensor).  adj_freq may be NULL; otherwise it
 * receives compute_matching_loci's returned (possibly inverted) frequencies:
 * adj_freq[(set * 2 + 0) * n_sites + site] = ref_freq, [(set * 2 + 1) * n_sites + site] = tgt_freq. */
int sai_site_flags(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                   const uint32_t* counts, int32_t n_sets, const sai_params* sets_host,
                   double* tgt_freq, uint64_t* planes, int64_t plane_stride, double* adj_freq, void* stream);

/* Kernel 3: window -> site-index range.  Replaces the per-window position masks of
 * WindowGenerator._window_generator (window_generator.py:173-183) for a resident block with
 * non-decreasing positions: lo[w] = first site with pos >= win_start[w], hi[w] = one past the
 * last site with pos <= win_end[w] (inclusive windows, utils.py:607-610). */
int sai_window_bounds(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                      const int64_t* win_start, const int64_t* win_end, int32_t* lo, int32_t* hi,
                      void* stream);

/* The same search for a block that holds several chromosome pieces back to back (a rank's share of
 * a whole-genome window list, chunk_generator.py:111-142 applied to the concatenated list): window
 * w is searched only inside sites [seg_lo[w], seg_hi[w]) of the block -- positions ascend inside a
 * piece, not across pieces -- and lo/hi are block-relative like sai_window_bounds'.  Segment bounds
 * are clamped into [0, n_sites]. */
int sai_window_bounds_seg(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                          const int64_t* win_start, const int64_t* win_end, const int32_t* seg_lo,
                          const int32_t* seg_hi, int32_t* lo, int32_t* hi, void* stream);

/* Kernel 4 (four launches): one record per (set, window): U count (u_statistic.py:92-96: condition
 * sites whose effective target frequency is > sets_host[set].x),
 * numpy 'linear' nanquantile of the effective target frequency over condition sites
 * (q_statistic.py:92-100), and both candidate lists (u_statistic.py:95, q_statistic.py:101) in
 * ascending site order, laid out as a CSR in (set, window) order -- deterministic, so 1-GPU and
 * sharded runs produce identical bytes.
 *   records[set * n_windows + w]
 *   cdd_off[(set * n_windows + w) * 2 + {0,1}] = start of the record's U / Q list inside
 *     cdd_u / cdd_q = exclusive prefix sum of u_count / n_cdd_q over the preceding records
 *     (-1 when the list would end beyond the buffer's capacity; such lists are not written);
 *   list entries are pos[site] when `pos` is non-NULL, else block-relative site indices;
 *   cdd_total[0..1] = entries needed for all U / Q lists: when a total exceeds its capacity,
 *     re-run with larger buffers.  cdd_total must hold sai_window_total_words(n_sets, n_windows)
 *     int64 words: the two totals, then scratch of the parallel prefix sum (one pair per 1024 records).
 * `quantile` and `x` are taken from sets_host[set]; whether the rows carry inverted words from the
 * sets' anc_allele_available, exactly as the call that wrote them decided it -- so sets_host must be
 * the array that call was given.  `planes` / `plane_stride`: the rows of these n_sets sets (plane_stride
 * words per tile, this call's "any" word at word 0 of the pointer).  tgt_freq is read only at sites
 * whose condition bit is set, at the slot the "any" word gives them. */
int64_t sai_window_total_words(int32_t n_sets, int32_t n_windows); /* -1 on bad arguments */
int sai_window_stats(sai_ctx* ctx, int64_t n_sites, const double* tgt_freq, const uint64_t* planes,
                     int64_t plane_stride, int32_t n_sets, const sai_params* sets_host, int32_t n_windows,
                     const int32_t* lo, const int32_t* hi, const int32_t* pos,
                     sai_window_record* records, int64_t* cdd_off, int32_t* cdd_u, int64_t cap_u,
                     int32_t* cdd_q, int64_t cap_q, int64_t* cdd_total, void* stream);

/* One window per call -- the whole of UStatistic.compute (u_statistic.py:79-99) and
 * QStatistic.compute (q_statistic.py:79-104) for the plugin classes: the fused site pass over the
 * window's blocks (`pops` = ref, tgt, sources as DEVICE tiled blocks of exactly the window's
 * n_sites sites), the window statistics over all of them, and the results in the caller's HOST
 * buffers: *record_host, and the U / Q candidates as 0-based site indices (ascending) in
 * cdd_u_host / cdd_q_host, each with room for n_sites entries (record_host->u_count and
 * ->n_cdd_q say how many were written).  Unlike the other entry points this one SYNCHRONISES
 * `stream` before it returns and keeps its device and pinned-host scratch in the ctx (grown on
 * demand, released by sai_ctx_destroy).  n_sites == 0 gives U = 0 and Q = NaN. */
int sai_single_window(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                      const sai_params* set_host, sai_window_record* record_host, int32_t* cdd_u_host,
                      int32_t* cdd_q_host, void* stream);

/* ---- prepared launch sequences ------------------------------------------------------------ */

/* The arguments of a resident block's site pass and windows stage do not change from step to step
 * (ChunkPreprocessor.run over a resident region, chunk_preprocessor.py:105-147; one statistic pass per
 * parameter sweep).  A plan stores copies of the arguments of the entry points above -- parameter
 * sets and population descriptors included; device and pinned-host pointers are the caller's and
 * must stay valid -- and sai_plan_run replays them in the order they were added, on `stream`, with
 * the same validation and the same kernels as the direct calls.  One call from the host language
 * per sequence instead of one per kernel: a step of a small block costs the host less than the GPU.
 * A plan belongs to its ctx; it is not thread-safe. */
typedef struct sai_plan sai_plan;
int sai_plan_create(sai_ctx* ctx, sai_plan** plan_out);
int sai_plan_destroy(sai_plan* plan);
int sai_plan_run(sai_plan* plan, void* stream);
/* sai_site_counts */
int sai_plan_add_site_counts(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts);
/* sai_site_pass (packed2 == 0) or sai_site_pass_packed2 (packed2 != 0) */
int sai_plan_add_site_pass(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                           uint32_t* counts, int32_t n_sets, const sai_params* sets_host,
                           int32_t freq_mode, double* tgt_freq, uint64_t* planes, int64_t plane_stride,
                           int32_t packed2);
/* sai_site_flags without adj_freq */
int sai_plan_add_site_flags(sai_plan* plan, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                            const uint32_t* counts, int32_t n_sets, const sai_params* sets_host,
                            double* tgt_freq, uint64_t* planes, int64_t plane_stride);
/* sai_window_bounds (seg_lo == NULL) or sai_window_bounds_seg */
int sai_plan_add_window_bounds(sai_plan* plan, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                               const int64_t* win_start, const int64_t* win_end, const int32_t* seg_lo,
                               const int32_t* seg_hi, int32_t* lo, int32_t* hi);
/* sai_window_stats */
int sai_plan_add_window_stats(sai_plan* plan, int64_t n_sites, const double* tgt_freq, const uint64_t* planes,
                              int64_t plane_stride, int32_t n_sets, const sai_params* sets_host,
                              int32_t n_windows, const int32_t* lo, const int32_t* hi, const int32_t* pos,
                              sai_window_record* records, int64_t* cdd_off, int32_t* cdd_u, int64_t cap_u,
                              int32_t* cdd_q, int64_t cap_q, int64_t* cdd_total);
/* hipMemcpyAsync of n_bytes from device memory to (pinned) host memory: the records of a step */
int sai_plan_add_copy_to_host(sai_plan* plan, void* dst_host, const void* src, int64_t n_bytes);

/* Events carried BY a launch.  Between two consecutive site passes of one queue every marker a caller records
 * (the end of a pass for timing, the hand-over the host waits for before it enqueues the windows stage, the
 * start of the next pass) is a packet of its own that the queue works off before the next dispatch: ~10 us
 * for three of them, an eighth of a short pass (C2: 75 us; nothing in the reference corresponds -- its
 * statistics run where the data is, sai.py:146-151).  A launch can carry its events in its own dispatch
 * packet instead (hipExtLaunchKernelGGL): `start` is stamped when the kernel begins, `stop` when it ends,
 * and no packet stands between this pass and the next.  sai_plan_set_pass_events attaches two such events
 * (either may be NULL) to the site pass of a plan (sai_plan_add_site_pass / _site_counts): every
 * sai_plan_run from then on stamps them; the caller waits for `stop` (sai_event_synchronize / _query) and
 * reads the pass's duration with sai_event_elapsed_ms.  Events belong to the ctx's device. */
int sai_event_create(sai_ctx* ctx, void** event_out);
int sai_event_destroy(void* event);
int sai_event_synchronize(void* event);
int sai_event_query(void* event, int32_t* done_out);
int sai_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out);
int sai_plan_set_pass_events(sai_plan* plan, void* start_event, void* stop_event);

/* ---- ABBA-BABA family: fd, df, Danc, Dplus --------------------- */

/* calc_freq's f64 division for every population of a counts tensor (stat_utils.py:48-52;
 * what calc_four_pops_freq, stat_utils.py:209-217, asks for): freqs[p * n_sites + site], NaN
 * where no individual is called.  Up to 9 populations (ref, tgt, 6 sources, outgroup). */
int sai_site_freqs(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                   const uint32_t* counts, double* freqs, void* stream);

/* Per (window, source): the pattern sums of calc_pattern_sum (stat_utils.py:220-272) for abba,
 * baba, bbaa, baaa, abaa and fd's two denominators -- products formed in population order, sums
 * taken in numpy's np.sum order, so they are the reference's doubles bit for bit -- and the four
 * statistics formed from them (fd_statistic.py:85-88, df_statistic.py:79-82,
 * danc_statistic.py:78-81, dplus_statistic.py:81-84; NaN on a zero denominator).
 * freqs = [ref, tgt, src_0..src_{n_src-1} (, outgroup)] as written by sai_site_freqs; without an
 * outgroup its frequency is 0 everywhere (stat_utils.py:213-214).
 * sums[(w * n_src + s) * 7 + k], stats[(w * n_src + s) * 4 + {fd, df, Danc, Dplus}]. */
int sai_window_fourpop(sai_ctx* ctx, int64_t n_sites, int32_t n_src, int32_t has_outgroup,
                       const double* freqs, int32_t n_windows, const int32_t* lo, const int32_t* hi,
                       double* sums, double* stats, void* stream);

/* calc_pattern_sum (stat_utils.py:220-272) for one arbitrary pattern over whole arrays: bit k of
 * pattern_bits set = population k (0 ref, 1 tgt, 2 src, 3 out) contributes its frequency ('b'),
 * clear = one minus it ('a'); *sum_out (device) = np.sum of the per-site products, in numpy's
 * order.  One wavefront; the statistic classes use sai_window_fourpop instead. */
int sai_pattern_sum(sai_ctx* ctx, int64_t n_sites, const double* ref_freq, const double* tgt_freq,
                    const double* src_freq, const double* out_freq, int32_t pattern_bits, double* sum_out,
                    void* stream);

/* ---- DD --------------------------------------------------------- */

/* Per site and source individual a: out[a * n_sites + site] = sum over the individuals b of `pop`
 * of |src[a][site] - pop[b][site]| on the raw int8 dosages (negative = missing enters as that
 * number): the per-site terms of scipy's cdist(src.T, pop.T, "cityblock") (dd_statistic.py:64-66).
 * One streaming pass over `pop` per two source individuals. */
int sai_site_absdiff(sai_ctx* ctx, int64_t n_sites, const sai_pop* pop, const sai_pop* src,
                     uint32_t* out, void* stream);

/* Per window the DD value of one source population (dd_statistic.py:68-74): with
 * T_ref[a] / T_tgt[a] the window sums of the per-site terms above,
 * dd = mean_a( T_ref[a] / n_ref_ind - T_tgt[a] / n_tgt_ind ), summed in numpy's order.
 * scratch: n_windows * n_src_ind doubles. */
int sai_window_dd(sai_ctx* ctx, int64_t n_sites, int32_t n_src_ind, const uint32_t* ad_ref,
                  int32_t n_ref_ind, const uint32_t* ad_tgt, int32_t n_tgt_ind, int32_t n_windows,
                  const int32_t* lo, const int32_t* hi, double* scratch, double* dd, void* stream);

/* The same terms from the site pass itself: sai_site_pass with DD riding along, so that the genotypes of
 * ref and tgt are read ONCE for the counts, the per-site decision and DD (the stand-alone form above streams a
 * population again per two source individuals).  dd->first_pop .. first_pop + n_pops - 1 name the source
 * populations of `pops` whose individuals are DD's sources (all of them together at most SAI_DD_FUSED_ROWS
 * rows); dd->absdiff receives [2][n_rows][n_sites] uint32: [0] against pops[0] (ref), [1] against pops[1]
 * (tgt), row r = the r-th individual of those populations in order -- the slices sai_window_dd takes.
 * n_sets == 0 asks for the counts (and DD) only: every population behind tgt is then just counted (an
 * outgroup may ride along); with parameter sets they are the sources of the decision, as in sai_site_pass.
 * counts / tgt_freq / planes as in sai_site_pass.  SAI_ERR_UNSUPPORTED -- more source individuals -- tells the
 * caller to take sai_site_pass + sai_site_absdiff. */
#define SAI_DD_FUSED_ROWS 4
typedef struct sai_dd_rows {
  int32_t first_pop;
  int32_t n_pops;
  uint32_t* absdiff;
} sai_dd_rows;
int sai_site_pass_dd(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                     int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                     uint64_t* planes, int64_t plane_stride, const sai_dd_rows* dd, void* stream);
/* as a step of a prepared launch sequence (sai_plan_add_site_pass) */
int sai_plan_add_site_pass_dd(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                              int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                              uint64_t* planes, int64_t plane_stride, const sai_dd_rows* dd);

/* ---- packed2: optional 2-bit layout ----------------------------- */

/* For dosages in {0, 1, 2} plus missing (unphased diploid or haploid biallelic calls) a block can
 * be held 4x denser: 2 bits per individual (0, 1, 2 = dosage, 3 = missing).  Per tile of 64
 * consecutive sites the individuals form n_full = n_ind / 64 full groups of 64 (16 bytes per site:
 * a 1 KiB block, site-major) and one tail group of the remaining n_ind % 64 in
 * w_tail = ceil((n_ind % 64) / 16) words per site (a 256 * w_tail byte block), tiles back to back;
 * with W = n_full * 256 + w_tail * 64 words per tile:
 *   field(site, ind) = bits [2 * (ind % 16), +2) of uint32 word
 *       (site / 64) * W + (ind / 64) * 256 + (site % 64) * 4 + (ind % 64) / 16         (full group)
 *       (site / 64) * W + n_full * 256 + (site % 64) * w_tail + (ind % 64) / 16        (tail group)
 * Padding individuals hold code 0, padding sites of the last tile code 3.  The byte count of a
 * launch is 4x smaller, so throughput figures on this layout are always reported separately from
 * the int8 ones. */
int64_t sai_packed2_bytes(int64_t n_sites, int32_t n_ind); /* -1 on bad arguments */

/* Convert a tiled int8 block; *n_unrepresentable (device) receives the number of words holding a
 * dosage above 2 -- the caller must not use the packed block when it is non-zero. */
int sai_pack2_from_tiles(sai_ctx* ctx, const int8_t* tiles, int64_t n_sites, int32_t n_ind,
                         uint8_t* packed, int32_t* n_unrepresentable, void* stream);

/* sai_site_pass on packed2 blocks (pops[p].tiles points at the packed block); n_sets may be 0 to
 * obtain only the counts.  Results are identical to the int8 entry points. */
int sai_site_pass_packed2(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                          uint32_t* counts, int32_t n_sets, const sai_params* sets_host,
                          int32_t freq_mode, double* tgt_freq, uint64_t* planes, int64_t plane_stride,
                          void* stream);

/* ---- synthetic data ("synth-v1") ---------------------------------- */

/* Counter-based generator: every byte is a pure function of (seed, chrom, site, population
 * stream, individual), so any shard on any GPU and the host produce identical data.
 * pop_stream: 0 = ref, 1 = tgt, 2.. = sources.  missing_per_million: probability * 1e6 that a
 * call is missing (-ploidy).  Fills the tiled SoA block for sites [site0, site0 + n_sites). */
int sai_synth_fill(sai_ctx* ctx, uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites,
                   int32_t pop_stream, int32_t n_ind, int32_t ploidy, int32_t missing_per_million,
                   int8_t* tiles, void* stream);
/* Same bytes on the host, in reference order [site][individual] (row stride = n_ind). */
int sai_synth_fill_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites,
                        int32_t pop_stream, int32_t n_ind, int32_t ploidy,
                        int32_t missing_per_million, int8_t* site_major_host);
/* gaps_host[i] = pos[site0 + i] - pos[site0 + i - 1] (1..49); pos[-1] = 0. */
int sai_synth_gaps_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites,
                        int32_t* gaps_host);
int sai_synth_gaps(sai_ctx* ctx, uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites,
                   int32_t* gaps, void* stream);

/* ---- ingest (host side, no GPU involved) ------------------------------------------------- */

/* First and last POS of the first contiguous run of `chrom` records; -1/-1 when the chromosome
 * is absent.  Replaces the pysam scan of ChunkGenerator.__init__ (chunk_generator.py:64-73).
 * Plain, gzip and bgzip files are accepted. */
int sai_vcf_scan(const char* path, const char* chrom, int64_t* first_pos, int64_t* last_pos);

/* One pass over a VCF: for the records of `chrom` with start <= POS <= end (either bound -1 =
 * open), the unphased ALT dosage of the selected samples as int8 [record][sample] plus int32
 * positions.  Replaces read_geno_data + reshape_genotypes (utils.py:78-186, 389-410): first ALT
 * only, '.' allele = -1, each call padded / cut to ploidy[sample] alleles, alleles summed.  With
 * `anc_bed_path` (columns chrom, start, pos, allele) it also applies check_anc_allele
 * (utils.py:492-555): only listed sites are kept, sites whose ancestral allele is neither REF nor
 * ALT are dropped, and where ALT is ancestral every allele call a becomes |a - 1|.  Lines are
 * tokenised by `n_threads` threads.  The block is owned by the library until
 * sai_vcf_block_free. */
typedef struct sai_vcf_block sai_vcf_block;
int sai_vcf_load(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                 const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                 int32_t n_threads, sai_vcf_block** block_out);
/* n_records = rows held; n_matched = records of the chromosome/region before polarisation;
 * n_anc_entries = ancestral-allele entries found for the chromosome/region. */
int sai_vcf_block_info(const sai_vcf_block* block, int64_t* n_records, int64_t* n_matched,
                       int64_t* n_anc_entries);
int sai_vcf_block_copy(const sai_vcf_block* block, int32_t* pos_host, int8_t* dosage_host);
int sai_vcf_block_free(sai_vcf_block* block);

/* Streaming form of the ingest for the GPU tokenizer: a producer thread walks the file exactly as
 * sai_vcf_load does (plain / gzip / bgzip with parallel inflate, tabix seek, early stop after the
 * region) but only INDEXES the record lines -- chromosome and region filter, POS, the
 * ancestral-allele decision of check_anc_allele (utils.py:492-555: keep, flip or drop; it needs the
 * fixed columns only), the index of GT inside FORMAT, where the sample columns start -- and copies
 * the text as it is into the caller's two pinned buffers alternately; sai_tokenize_gt then turns the
 * text into dosages on the GPU.  sai_vcf_stream_next hands out batch k (it blocks until the batch is
 * ready) and releases the buffer of batch k-1, so the caller must have finished its H2D copy of
 * batch k-1 by then; the line arrays stay valid until the following call.  line_off = offset of the
 * first sample column inside the batch text, line_len = bytes from there to the end of the line
 * (without "\r").  *done = 1 (and no batch) once everything has been handed out, or the producer's
 * error as the status.  sai_vcf_stream_selection: slot_of_col[c] = output slot of VCF sample column
 * c or -1 (valid once a batch has been returned), n_matched / n_anc_entries as sai_vcf_block_info
 * (complete once done was reported). */
typedef struct sai_vcf_stream sai_vcf_stream;
int sai_vcf_stream_open(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                        const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                        int32_t n_threads, void* pinned0_host, void* pinned1_host, int64_t buffer_bytes,
                        sai_vcf_stream** stream_out);
int sai_vcf_stream_next(sai_vcf_stream* stream, int32_t* buffer_index, int64_t* n_text_bytes, int64_t* n_lines,
                        const int64_t** line_off_host, const int32_t** line_len_host, const int32_t** line_pos_host,
                        const uint8_t** line_flip_host, const uint8_t** line_gi_host, int32_t* done);
int sai_vcf_stream_selection(sai_vcf_stream* stream, int32_t* slot_of_col_host, int32_t capacity, int32_t* n_cols,
                             int64_t* n_matched, int64_t* n_anc_entries);
int sai_vcf_stream_close(sai_vcf_stream* stream);

/* The GPU half: the sample columns of n_lines record lines -> dosages.  text = the batch text in
 * HBM; line_off / line_len / line_flip / line_gi = the index of sai_vcf_stream_next (device copies);
 * slot_of_col (n_cols entries) and ploidy_of_slot (n_out entries) = the sample selection.  For every
 * selected column the GT sub-field is read exactly as read_geno_data + reshape_genotypes do
 * (utils.py:78-186, 389-410): alleles separated by | or /, '.' or an empty allele = -1, the call
 * padded / cut to the slot's ploidy, alleles summed; a flipped line stores sum |a - 1| instead
 * (utils.py:531-555).  out[(line * n_out + slot)] = int8 dosage, [record][sample] like
 * sai_vcf_block_copy.  status[line] = 0, or non-zero where the host reader would have refused the
 * line (unparsable genotype, too few sample columns, dosage outside int8): the caller then lets
 * sai_vcf_load produce the reference error text.  One wavefront per line. */
int sai_tokenize_gt(sai_ctx* ctx, const char* text, int64_t n_text_bytes, int64_t n_lines, const int64_t* line_off,
                    const int32_t* line_len, const uint8_t* line_flip, const uint8_t* line_gi, int32_t n_cols,
                    const int32_t* slot_of_col, int32_t n_out, const int32_t* ploidy_of_slot, int8_t* out,
                    int32_t* status, void* stream);

/* bgzip input inflated on the GPU.  A BGZF file (the container of every .vcf.gz that tabix can
 * index) is a sequence of independent gzip members of at most 64 KiB of text; sai_inflate_bgzf
 * inflates the members of a batch side by side, one wavefront per member (RFC 1951: stored, fixed
 * and dynamic blocks), so the compressed bytes cross PCIe instead of the text and the host's
 * inflate (vcf_ingest.cpp, libdeflate on the box's cores) is out of the way.  comp = the compressed
 * bytes of the batch in HBM (4-byte aligned, n_comp_bytes a multiple of 4: pad the tail);
 * members[m] = where member m's raw deflate stream lies in comp and where its text goes in `text`
 * (device copy of the table sai_bgzf_stream_next hands out); status[m] = 0, or non-zero for a member
 * that is not valid DEFLATE of exactly isize bytes, or whose text fails the CRC-32 of its trailer
 * (checked on the GPU as well, by a second launch); nothing outside [out_off, out_off + isize) is
 * ever written. */
typedef struct sai_bgzf_member {
  int64_t data_off;  /* first byte of the raw deflate stream inside the compressed batch */
  int64_t out_off;   /* first byte of the member's text inside the batch text */
  uint32_t data_len; /* compressed bytes */
  uint32_t isize;    /* uncompressed bytes (<= 65536) */
  uint32_t crc;      /* CRC-32 of the text (gzip trailer) */
  uint32_t reserved;
} sai_bgzf_member;
int sai_inflate_bgzf(sai_ctx* ctx, const void* comp, int64_t n_comp_bytes, const sai_bgzf_member* members,
                     int32_t n_members, void* text, int64_t n_text_bytes, int32_t* status, void* stream);

/* The host side of that path.  sai_bgzf_stream_open starts a reader thread that hands the file's
 * BGZF members over as they are -- whole members, padded to a multiple of 4 bytes -- in the caller's
 * two pinned buffers alternately, at most text_batch_bytes of text per batch; SAI_ERR_UNSUPPORTED
 * when the file is not bgzip; n_samples = 0 asks for the record index only (the positions of a
 * chromosome).  A region (start >= 0) of a file with a usable <vcf>.tbi is a seek: the reader hands
 * over only the members from the region's first record (the tabix linear index) to the member that
 * holds the first record of a later 16 kb window, as each worker of the reference reads only its own
 * region (utils.py:117-138, chunk_generator.py:130-142).  sai_bgzf_stream_region says what was
 * decided: *file_begin / *file_stop = compressed offsets of the first and last member read (-1 / -1:
 * the index holds no record for the region, nothing is read; file_stop -1: to the end of the file)
 * and *first_text_skip = bytes of the first member's text that precede the region's first record --
 * the caller starts its line table behind them.  sai_bgzf_stream_next returns batch k (blocking) and
 * releases the buffer of batch k-1 (its H2D copy must be over); members_host = the table for
 * sai_inflate_bgzf, valid until the following call.  The caller inflates the batch on the GPU behind
 * the n_carry bytes the previous batch left over (its last, incomplete line), copies carry + text to
 * the host once, and calls sai_vcf_index_text(text_host, n_bytes = n_carry + text, n_carry, the
 * batch's member table, or NULL / 0 when the CRCs have been checked on the GPU): the text of every
 * member is checked against its CRC-32, the header is consumed, the complete record lines are indexed exactly as sai_vcf_stream_next reports them
 * (offsets relative to text_host, i.e. to the same place in the device copy) and *n_usable = bytes up
 * to the end of the last complete line (is_last: all of it); the header itself was read when the
 * stream was opened, '#' lines are skipped.  *done = 1 once the region has been
 * passed (stop reading).  After the last batch the caller indexes the left-over carry with
 * n_members = 0, is_last = 1.  sai_bgzf_stream_selection as sai_vcf_stream_selection. */
typedef struct sai_bgzf_stream sai_bgzf_stream;
int sai_bgzf_stream_open(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                         const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                         int32_t n_threads, void* comp0_host, void* comp1_host, int64_t comp_buffer_bytes,
                         int64_t text_batch_bytes, sai_bgzf_stream** stream_out);
int sai_bgzf_stream_next(sai_bgzf_stream* stream, int32_t* buffer_index, int64_t* n_comp_bytes, int32_t* n_members,
                         const sai_bgzf_member** members_host, int64_t* n_text_bytes, int32_t* done);
int sai_bgzf_stream_region(sai_bgzf_stream* stream, int64_t* file_begin, int64_t* file_stop, int64_t* first_text_skip);
/* Early form of the release sai_bgzf_stream_next performs: the batch's compressed bytes have been
 * copied (and the member table too: it is refilled with the buffer), the reader may go on. */
int sai_bgzf_stream_release(sai_bgzf_stream* stream);
int sai_vcf_index_text(sai_bgzf_stream* stream, const char* text_host, int64_t n_bytes, int64_t n_carry,
                       const sai_bgzf_member* members_host, int32_t n_members, int32_t is_last, int64_t* n_usable,
                       int64_t* n_lines, const int64_t** line_off_host, const int32_t** line_len_host,
                       const int32_t** line_pos_host, const uint8_t** line_flip_host, const uint8_t** line_gi_host,
                       int32_t* done);
/* The same index from the line table of sai_text_line_starts / sai_text_line_heads (host copies):
 * heads_host[i * head_bytes ...] = the first bytes of line i, line_start_host[0 .. n_lines], the
 * line_info words; head_bytes must cover info[1] (the longest fixed-column part).  Record lines are
 * reported exactly as by sai_vcf_index_text, with offsets relative to the text the table was made of. */
int sai_vcf_index_heads(sai_bgzf_stream* stream, const char* heads_host, int32_t head_bytes, const int64_t* line_start_host,
                        const int32_t* line_info_host, int64_t n_lines, int64_t* n_record_lines,
                        const int64_t** line_off_host, const int32_t** line_len_host, const int32_t** line_pos_host,
                        const uint8_t** line_flip_host, const uint8_t** line_gi_host, int32_t* done);
int sai_bgzf_stream_selection(sai_bgzf_stream* stream, int32_t* slot_of_col_host, int32_t capacity, int32_t* n_cols,
                              int64_t* n_matched, int64_t* n_anc_entries);
int sai_bgzf_stream_close(sai_bgzf_stream* stream);

/* The line structure of a text batch in HBM (the text sai_inflate_bgzf produced), so that the host
 * can index the records from a few MB instead of the whole text.  sai_text_line_starts: n = number
 * of newlines (info[0]); line_start[i] = offset of line i for i = 0 .. n (line_start[0] = 0,
 * line_start[n] = first byte behind the last complete line = the bytes usable in this batch);
 * line_info[i] = bytes of line i up to and including its ninth tab -- the fixed columns CHROM ..
 * FORMAT -- or 1 for a '#' line, or (bytes of the line + 1) when it has fewer than ten columns, or
 * 4097 when the ninth tab is further than 4096 bytes away; bit 31 = the line ends with "\r\n";
 * info[1] = the largest of them; info[2] = 1 when line_capacity was too small (nothing beyond it is
 * written).  block_scratch: (n_bytes + 15) / 4096 + 2 int32.  The text buffer must be readable up to
 * the next 16-byte boundary behind n_bytes (aligned 16-byte loads).  sai_text_line_heads then copies
 * the first head_bytes (a multiple of 4) of lines 0 .. n_lines-1 into heads[line][head_bytes], padded
 * with '\n' behind the line's own newline. */
int sai_text_line_starts(sai_ctx* ctx, const char* text, int64_t n_bytes, int64_t line_capacity, int64_t* line_start,
                         int32_t* line_info, int32_t* block_scratch, int32_t* info, void* stream);
int sai_text_line_heads(sai_ctx* ctx, const char* text, int64_t n_bytes, const int64_t* line_start, int64_t n_lines,
                        int32_t head_bytes, void* heads, void* stream);

/* ---- output text (host side) -------------------------------------------------------------- */

/* The rows FeaturePreprocessor.process_items writes (feature_preprocessor.py:193-258), formatted
 * straight from the numeric window results: one TSV row per window
 *     chr \t start \t end \t <pop_columns> \t nsnps \t col_0 \t ... \n
 * where pop_columns is the caller's "ref\ttgt\tsrc1,src2\tout" and column c of window w is read at
 * cols[c].data + w * cols[c].stride_bytes as an int32 or a double (U's count and Q's value inside the
 * 24-byte records, the f64 blocks of the ABBA-BABA family and DD); a window with nsnps == 0 prints
 * "nan" in every column.  Numbers print as Python's str() prints them (shortest round-trip digits,
 * fixed notation for 1e-4 <= |x| < 1e16, "nan").  sai_format_log_rows writes the .U.log / .Q.log
 * rows "chr \t start \t end \t chr:pos,chr:pos,...|NA \n" from a CSR list: count of window w at
 * counts + w * count_stride_bytes (int32), its first entry at positions[offsets[w * offset_stride_words]].
 * The text is owned by the library until sai_text_free. */
enum sai_text_kind { SAI_TEXT_I32 = 0, SAI_TEXT_F64 = 1 };
typedef struct sai_text_column {
  const void* data;
  int64_t stride_bytes;
  int32_t kind;
  int32_t reserved;
} sai_text_column;
typedef struct sai_text sai_text;
int sai_format_score_rows(const char* chr_name_host, const char* pop_columns_host, int32_t n_windows,
                          const int64_t* windows_host, const int32_t* nsnps_host, int32_t n_cols,
                          const sai_text_column* cols_host, sai_text** text_out);
int sai_format_log_rows(const char* chr_name_host, int32_t n_windows, const int64_t* windows_host,
                        const void* counts_host, int64_t count_stride_bytes, const int64_t* offsets_host,
                        int64_t offset_stride_words, const void* positions_host, int32_t position_bytes,
                        sai_text** text_out);
int sai_format_doubles(const double* values_host, int64_t n, sai_text** text_out); /* one str(x) per line */
const char* sai_text_data(const sai_text* text, int64_t* n_bytes);
int sai_text_free(sai_text* text);

/* The same rows written straight to open files: the TSV rows of sai_format_score_rows go to tsv_fd and
 * the rows of sai_format_log_rows for each of the n_logs candidate lists to its own fd (any fd < 0 =
 * that output is not wanted).  ONE fan-out over the windows formats every output of a piece, and the
 * pieces are handed to writev() in window order -- no joined copy, no text crossing into the caller's
 * language (10^4 windows of C3, TSV + .U.log + .Q.log: 1.2 ms through the three sai_format_* calls, a
 * third of that here).  The files are the caller's (opened for appending, unbuffered); bytes_out, if not
 * NULL, receives the 1 + n_logs byte counts.  A failed write is SAI_ERR_ARG with errno's text (the fd is the caller's argument); what was
 * written before it stays in the files, as with process_items' row-by-row writes. */
#define SAI_MAX_LOGS 8
typedef struct sai_log_rows {
  const void* counts_host;     /* int32 at counts + w * count_stride_bytes */
  int64_t count_stride_bytes;
  const int64_t* offsets_host; /* first entry of window w at positions[offsets[w * offset_stride_words]] */
  int64_t offset_stride_words;
  const void* positions_host;  /* int32 or int64 (position_bytes); may be NULL when every count is 0 */
  int32_t position_bytes;
  int32_t fd;
} sai_log_rows;
int sai_write_window_rows(const char* chr_name_host, const char* pop_columns_host, int32_t n_windows,
                          const int64_t* windows_host, const int32_t* nsnps_host, int32_t n_cols,
                          const sai_text_column* cols_host, int32_t tsv_fd, int32_t n_logs,
                          const sai_log_rows* logs_host, int64_t* bytes_out);

/* In-memory counterpart of the ingest: narrow a reference-style [rows][cols] integer matrix (the
 * reference holds genotypes as int64 after utils.py:410) to the int8 the device layout uses, in one
 * multithreaded pass.  itemsize in {1, 2, 4, 8}; row_stride_bytes between consecutive rows, the
 * elements of a row contiguous.  Values below -128 can only be missing calls for calc_freq and are
 * stored as -128; a value above 127 is SAI_ERR_UNSUPPORTED. */
int sai_narrow_to_int8(const void* src, int32_t itemsize, int32_t is_signed, int64_t n_rows, int64_t n_cols,
                       int64_t row_stride_bytes, int8_t* dst, int32_t n_threads);

/* ---- measurement aid -------------------------------------------------------------------- */

/* Plain streaming read of `n_bytes` (multiple of 16) with 16-byte non-temporal loads, one wave per
 * contiguous 125 KiB run; the XOR of all 32-bit words is XOR-ed INTO *xor_out (device; the caller
 * zeroes it, so that a timed region holds this one kernel only).  No reference counterpart: bench.py times it to obtain the on-box read
 * ceiling that the site_counts rate is compared with, next to the 8 TB/s datasheet peak. */
int sai_probe_stream_read(sai_ctx* ctx, const void* buf, int64_t n_bytes, uint32_t* xor_out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAIHIP_H */
