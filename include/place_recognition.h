/* place_recognition.h — C ABI of libpr_amd.so, the MI355X (gfx950) implementation of the
 * so_dso_place_recognition hot path: generate_signatures (Scan-Context + M2DP) and match_signatures.
 *
 * The reference has no FFI; its boundary is two C++ classes, two executables and three MATLAB functions
 * (SURVEY.md §8-b).  Every entry point below names the reference interface it replaces (file:line under
 * /root/reference/place_recognition/).  Conventions: flat arrays, caller-owned buffers, int status
 * (0 = PR_OK, <0 = error, text via pr_last_error), no exceptions across the boundary, one host thread per
 * context, one HIP device + one HIP stream per context.  There is NO CPU fallback: every call fails with
 * PR_EHIP when no gfx950 device is usable.
 */
#ifndef PLACE_RECOGNITION_H
#define PLACE_RECOGNITION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pr_ctx pr_ctx;
typedef struct pr_sigset pr_sigset;
typedef struct pr_clouds pr_clouds;

enum { PR_OK = 0, PR_EINVAL = -1, PR_ENOMEM = -2, PR_EHIP = -3, PR_EIO = -4, PR_ENAN = -5 };
enum { PR_TYPE_SC = 0, PR_TYPE_M2DP = 1, PR_TYPE_DELIGHT = 2, PR_TYPE_GIST = 3, PR_TYPE_BOW = 4 };   /* run_test.m:26-36 `type` */
enum { PR_ROLE_QUERY = 0, PR_ROLE_DB = 1 };         /* hist1 / hist2 of run_test.m:1 */
enum { PR_F64 = 0, PR_F32 = 1 };
enum { PR_HOST = 0, PR_DEVICE = 1 };
/* arithmetic of the SC and M2DP matchers (processSC.m:22-33, processM2DP.m:12-22 on the GPU): split-f16 (fp32 operands carried
 * as f16 hi + lo, three f16 MFMAs per product, fp32 accumulate; default, 2-3x faster, same 1e-7 error as fp32), plain fp32 MFMA, or
 * PR_SC_ARITH_F16 = BASELINE.json config 5's "fp16 descriptors": spectra / rows stored as ONE f16 (2976 B per SC entry and channel),
 * one f16 MFMA per product, fp32 accumulate.  No reference counterpart (run_test.m handles fp64 only).  Its all-pairs distances are
 * within PR_F16_DISTANCE_BOUND of the exact ones (worst case; ~1.3e-4 observed; SURVEY.md asks for 1e-3 on typical data); the top-k
 * calls keep EXACT indices: the k + 56 best of the f16 pass are re-evaluated in fp64, and a query whose candidate list does not provably
 * contain the exact top-k, or whose re-evaluated candidates could change places under the sigma error of the f16 pass (pr_f16_margin_dev),
 * is recomputed in split-f16 (host calls: automatically, PR_WARN_F16_FALLBACK is raised) */
#define PR_F16_DISTANCE_BOUND 2e-3   /* |d_f16 - d| per channel: 8 u (u = 2^-11) on the correlation of unit-norm rows, halved */
/* Order check: the relative error allowed for a row sigma of the all-pairs pass is eps = SIGMA_REL + DIST_ERR / sigma.  SIGMA_REL: the
 * systematic part (the minimum over 120 noisy variants compresses a row by 2.5e-7 relative in the fp32-grade passes, whatever the row length).
 * DIST_ERR: the largest error of a single distance - |sigma(d + e) - sigma(d)| <= max |e| holds for ANY error pattern (sigma is 1-Lipschitz in
 * the sup norm), also when the errors of many entries coincide, as they do for near-copies of one place (round 5; before, the term was
 * 4 x rms noise / (sigma sqrt(n - 1)), which assumes independent errors: tools/fuzz_all.py, seed 13 case 21, a row of 168 entries of which 99
 * are near-copies of two places, returned ranks 2 / 3 swapped in PR_SC_ARITH_F16 without a flag). */
#define PR_F16_SIGMA_REL 2e-4
#define PR_F16_DIST_ERR 2e-4         /* observed < 1.3e-4 (the worst-case bound of the arithmetic is PR_F16_DISTANCE_BOUND) */
#define PR_F32_SIGMA_REL 5e-7
#define PR_F32_DIST_ERR 2e-7         /* split-f16 / fp32 passes: 2.5e-8 rms, 1.3e-7 at most (tools/probe_bias.py, n = 20 000) */
enum { PR_SC_ARITH_F16X2 = 0, PR_SC_ARITH_F32 = 1, PR_SC_ARITH_F16 = 2 };
/* What a zero-norm SC row does.  MATLAB divides 0/0 (processSC.m:16,19): every distance to or from that signature is NaN, and
 * normalize(.,2) / min (run_test.m:40,57) leave NaNs out [normalize's 'omitnan' from memory], so the signature simply never matches.
 * PR_NAN_EXCLUDE (default) does exactly that and reports PR_WARN_NAN_ROWS; PR_NAN_FAIL turns it into the error PR_ENAN at pr_sync. */
enum { PR_NAN_EXCLUDE = 0, PR_NAN_FAIL = 1 };
enum { PR_WARN_NAN_ROWS = 1, PR_WARN_M2DP_SVD = 2, PR_WARN_F16_FALLBACK = 4, PR_WARN_ORDER_RESOLVED = 8,
       PR_WARN_ORDER_UNRESOLVED = 16 };   /* bits of pr_take_warnings; ORDER_RESOLVED: a query was answered from its exact fp64 row (order or containment
                                            check, below); the last one: a caller of the sharded per-pass form stopped (last_pass != 0) with flagged queries left -
                                            those keep the answer of the re-evaluated candidate list.  The library's own calls, stream-ordered or not, run every pass */

#define PR_SC_SIG_LEN 2400    /* 2 x numS*numR = 2 x 60*20, SC/SC.h:7-8, test_sc.cpp:37-38 */
#define PR_M2DP_SIG_LEN 384   /* 2 x (numP*numQ + numS*numR) = 2 x 192, M2DP/M2DP.h:7-10, test_m2dp.cpp:37-39 */
#define PR_DELIGHT_SIG_LEN 256 /* BINS, DELIGHT/DELIGHT.h:9; 16 rows (histograms) per signature, test_delight.cpp:36-37 */

/* ---- context ------------------------------------------------------------------------------------ */
int pr_create(int device_id, pr_ctx** out);
/* The same on a stream the CALLER owns (a hipStream_t, e.g. the stream a torch / RCCL process group enqueues on): every kernel of
 * the context is ordered with the caller's work on that stream, no host synchronisation is needed between the library's phases and
 * the caller's collectives (SURVEY.md §8-e "issue A and B on the compute stream").  The stream must outlive the context. */
int pr_create_on_stream(int device_id, void* hip_stream, pr_ctx** out);
void pr_destroy(pr_ctx* ctx);
const char* pr_last_error(const pr_ctx* ctx);        /* valid until the next call on ctx; ctx may be NULL */
const char* pr_version(void);
/* Selects the matcher arithmetic (SC and M2DP) for signature sets created AFTERWARDS (a set is packed for one arithmetic; matching two
 * sets packed differently is PR_EINVAL).  Initial value: PR_SC_ARITH_F16X2, or PR_SC_ARITH_F32 if the environment has
 * PR_SC_MATCH=f32. */
int pr_set_sc_arith(pr_ctx* ctx, int arith);
int pr_get_sc_arith(const pr_ctx* ctx);
int pr_sync(pr_ctx* ctx);                            /* waits for the context's stream; reports deferred errors */
int pr_set_nan_policy(pr_ctx* ctx, int policy);      /* PR_NAN_EXCLUDE | PR_NAN_FAIL */
int pr_get_nan_policy(const pr_ctx* ctx);
/* on != 0: EVERY query of a top-k call is treated as flagged, i.e. answered from its exact fp64 row (DESIGN.md section 2 "Returned order"):
 * returned scores are then the reference's doubles to rounding (|score - oracle| < 1e-9 whatever |z|; by default they carry the fp32 pass's
 * ~2e-7 relative error of the row sigma, 3e-5 absolute at z = -160).  The host calls, pr_group and stream-ordered calls resolve all queries (passes of 64;
 * a stream-ordered call of m queries chains ceil(m / 64) of them).  Off by default; the environment variable PR_FORCE_ORDER_FLAGS=1 sets it
 * at creation (tests). */
int pr_set_exact_statistics(pr_ctx* ctx, int on);
/* Binary intensity channel (no reference counterpart as a switch; the arithmetic is processSC.m:15-33 on the values SC/SC.cpp:67-72 writes:
 * channel 1 of an SC signature is 0 / 1).  In PR_SC_ARITH_F16X2 the pack notices whether every channel-1 row of a set has all of its non-zero
 * entries equal and positive; the normalised row is then 1/sqrt(ones) on `ones` bins and every one of the 120 products of processSC.m:30 is
 * count / sqrt(ones_q ones_d) with an INTEGER count.  pr_distances_dev / the top-k calls then compute channel 1 with ONE f16 product per
 * term on the hi halves of the same packed images and round max x sqrt(ones_q ones_d) to the nearest integer: the distance is exact (to the
 * fp32 rounding of the final expression, ~1e-7) whenever |x - rint(x)| + b < 1 for the pair, x the computed count of its best variant and b a
 * rigorous bound of the pass's error evaluated ON THE DEVICE from statistics of the two sets (rounding residual norms of the packed spectra) and
 * the pair's ones; the kernel tests every pair.  Sets with non-binary rows or too many ones (beyond ~550 per signature), or a call in which a
 * pair fails the test, get channel 1 from the split-f16 kernel as channel 0 does - same results to 1e-7, no host round trip either way.  on = 0 always takes the split-f16 kernel for both channels (environment: PR_SC_BINARY=0). */
int pr_set_sc_binary(pr_ctx* ctx, int on);
/* *state = 1 when a pr_distances_dev call on these two packed sets takes the binary path for channel 1, 2 when it did and the last call's
 * per-pair rounding test failed somewhere (channel 1 was then redone in split-f16), 0 when the bound rules the path out (reads the sets'
 * statistics back: synchronises; the device takes its own decision from the same numbers). */
int pr_sc_binary_state(pr_ctx* ctx, const pr_sigset* q, const pr_sigset* db, int32_t* state);
/* Measurement hooks (bench.py): on != 0 records HIP events on the context's stream around the matcher launches of every pr_distances_dev call;
 * pr_last_distance_timing waits for the last call's events and returns ms[3] = {channel-0 (or the only) launch, channel-1 single-product
 * launch, channel-1 split-f16 launch}; of the two channel-1 launches one has left at once. */
int pr_set_kernel_timing(pr_ctx* ctx, int on);
int pr_last_distance_timing(pr_ctx* ctx, float* ms);
int pr_take_warnings(pr_ctx* ctx);                   /* PR_WARN_* bits raised since the last call (synchronises the stream), then cleared */
void* pr_stream(pr_ctx* ctx);                        /* the context's hipStream_t (for event timing by the caller) */

/* ---- host-buffer entry points = the reference's own call boundary --------------------------------- */

/* Replaces SC::getSignature looped as in SC/test_sc.cpp:40-56 (SC/SC.h:10-23, SC/SC.cpp:12-76; PCA alignment
 * utils/pts_align.h:7-46 happens inside, as in SC.cpp:17).  Clouds in CSR layout: xyz[offs[N]][3] f64 camera
 * frame, inten[offs[N]] f32, offs[N+1].  out[N][2400] = [structure | intensity], bin = sector*20 + ring. */
int pr_sc_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                   double max_rho, double* out);

/* Replaces the per-cloud body of M2DP/test_m2dp.cpp:41-68: align_points_PCA once, 4 sign variants,
 * M2DP::getSignature (M2DP/M2DP.h:12-30, M2DP/M2DP.cpp:38-109) each.  out[4N][384]. */
int pr_m2dp_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                     double max_rho, double* out);
/* The rows (cloud * 4 + variant, ascending) of the LAST pr_m2dp_generate / pr_m2dp_generate_dev call of this context whose leading
 * singular pair is not unique (sigma_2 / sigma_1 > ~0.99: M2DP/M2DP.cpp:94-103's JacobiSVD returns whichever of the two near-equal
 * directions its sweeps end on, and so does this library - those rows may differ from the reference's; PR_WARN_M2DP_SVD is the
 * call-wide bit).  At most cap rows are written, *count is the number of such rows (the library records the first 1024 pairs of a call). */
int pr_m2dp_svd_rows(pr_ctx* ctx, int32_t* rows, int32_t cap, int32_t* count);

/* Replaces DELIGHT::getSignature looped as in DELIGHT/test_delight.cpp:41-56 (DELIGHT/DELIGHT.h:11-18, DELIGHT.cpp:8-24;
 * PCA alignment inside).  out[16N][256]: 16 intensity histograms per cloud. */
int pr_delight_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* out);

/* Replaces processDELIGHT(hist1, hist2) (match_signatures/processDELIGHT.m:1-38).  h1[16m][256], h2[16n][256];
 * dist: host f32 [m][n] (chi-square, min over the 4 octant permutations; +Inf when no bin is occupied). */
int pr_delight_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, float* dist);

/* Replaces processSC(hist1, hist2) (match_signatures/processSC.m:1-45).  h1[m][2400], h2[n][2400] host f64;
 * d_struct / d_int: host f32 [m][n], either may be NULL.  A zero-norm row (MATLAB: 0/0 = NaN, processSC.m:16,19) gives NaN distances and
 * PR_WARN_NAN_ROWS (default policy PR_NAN_EXCLUDE); with PR_NAN_FAIL the call returns PR_ENAN instead. */
int pr_sc_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n,
                   float* d_struct, float* d_int);

/* Replaces processM2DP(hist1, hist2) (match_signatures/processM2DP.m:1-22).  h1[4m][384], h2[4n][384]. */
int pr_m2dp_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n,
                     float* d_cnt, float* d_int);

/* Replaces run_test.m:26-57 (distance matrices, 2:1 z-score fusion :38-41, mask :47-53, row min :57),
 * generalised to top-k; k = 1 is the reference.  Ties -> lower index (MATLAB min).  idx[m][k] (0-based,
 * -1 when fewer than k candidates), score[m][k] fused z-score.  type = PR_TYPE_DELIGHT: no fusion (run_test.m:26-36),
 * score = the chi-square distance, h1[16m][256], h2[16n][256], p_weight ignored. */
int pr_match_topk(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n,
                  int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score);
/* The same with the scores as the reference holds them (MATLAB double, run_test.m:57 `diff_v`).  For SC and M2DP both variants
 * re-evaluate the k + 8 best pairs of the fp32 all-pairs pass in fp64 from the raw signatures (pr_rerank_dev): indices and scores
 * are those of the reference's double arithmetic given the row statistics (see DESIGN.md for the error of those). */
int pr_match_topk_f64(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n,
                      int32_t mask_width, double p_weight, int32_t k, int32_t* idx, double* score);

/* The two remaining `type`s of run_test.m:32-35, whose signatures have no fixed length (`cols` columns per row):
 *   gist: h [m][cols];            dist(i,j) = sum_c (h1[i,c] - h2[j,c])^2                    (processGIST.m:1-10)
 *   bow:  h [2 m][cols], rows alternate word ids | weights, padded with -1 (test_bow.cpp:147-162);
 *         dist(i,j) = 1 - DBoW2 L1 score                                                    (processBoW.m:1-38)
 * pr_match_topk_cols: mask + row minimum without fusion (run_test.m:47-57), type = PR_TYPE_GIST | PR_TYPE_BOW. */
int pr_gist_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, float* dist);
int pr_bow_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, float* dist);
int pr_match_topk_cols(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols,
                       int32_t mask_width, int32_t k, int32_t* idx, float* score);

/* BASELINE.json config 5, "fused SC + M2DP scoring" - NO reference counterpart (run_test.m handles one type per run);
 * build-defined as in SURVEY.md §6: score = [p z(sc_struct) + z(sc_int)] + [p z(m2dp_count) + z(m2dp_int)] with the row
 * z-scores of run_test.m:40, then mask and row minimum (run_test.m:47-57).  sc: [m][2400], m2dp: [4 m][384] of the same places.
 * pr_fuse_select2_dev is the device-level step (two channel pairs over the same grid, moments as pr_row_moments_dev). */
int pr_match_topk_fused(pr_ctx* ctx, const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2,
                        int32_t n, int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score);
int pr_match_topk_fused_f64(pr_ctx* ctx, const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2,
                            int32_t n, int32_t mask_width, double p_weight, int32_t k, int32_t* idx, double* score);
int pr_fuse_select2_dev(pr_ctx* ctx, const float* d_p, const float* d_i, const float* e_p, const float* e_i, int32_t m, int32_t n,
                        const double* mom_all, const double* mom2_all, int32_t G, int32_t q_row0, int32_t db_row0,
                        int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score);

/* ---- device-resident entry points (inputs already in HBM; what bench.py and the multi-GPU layer call) -- *
 * All are asynchronous on the context's stream; pr_sync() surfaces deferred errors (e.g. PR_ENAN).         */

/* A packed signature set: rows normalised as processSC.m:15-20 and stored as the per-ring sector spectra
 * (SC), or the 4-variant rows as-is (M2DP, processM2DP.m:15), in the MFMA operand layout of its role. */
#define PR_MAX_SIGS 4000000          /* capacity limit of one signature set (32-bit offsets inside the matchers) */
int pr_sigset_create(pr_ctx* ctx, int type, int role, int32_t max_sigs, pr_sigset** out);
void pr_sigset_destroy(pr_ctx* ctx, pr_sigset* s);
/* pr_sigset_pack inside a captured hipGraph: whether the image is zero-filled first is decided on the HOST from the set's history (a re-pack
 * of at least as many rows in the same geometry needs no fill) and baked into the graph as the presence or absence of a memset node.  A set
 * packed inside a captured graph must therefore not be re-packed with ANOTHER row count outside it between replays: the replayed pack would
 * run without the fill it then needs.  (Packing other sets, or this one with the captured count, is fine.) */
int pr_sigset_pack(pr_ctx* ctx, pr_sigset* s, const void* sig, int dtype, int where, int32_t n_sigs);
int32_t pr_sigset_count(const pr_sigset* s);
/* A DB that grows by one signature per keyframe (SC/test_sc.cpp:40-56 appends a row per cloud; run_test.m:57 matches a query against
 * everything before it).  The operand image is [channel][group][...]; pr_sigset_pack lays it out for the COUNT it is given, so a larger count
 * means a full re-pack.  pr_sigset_reserve fixes the layout at the CAPACITY (max_sigs of pr_sigset_create) instead: the set is emptied, and from
 * then on pr_sigset_pack (bulk: rows 0 .. n_sigs - 1, the set's new count) and pr_sigset_append (rows count .. count + n_new - 1, everything else
 * untouched) write into that one layout - an appended image is bit for bit the image a bulk pack of all rows writes, and the binary-channel
 * statistics of an SC set are folded in, not recomputed.  pr_sigset_append on an EMPTY set reserves by itself.  DB sets of SC
 * or M2DP signatures in the f16 arithmetics (the defaults); such a set is matched by the default kernels only (PR_EINVAL from
 * pr_distances_dev under PR_SC_KERNEL / PR_SC_ONLINE=h).  sig: n_new signatures, [n_new][2400] (SC) or [4 n_new][384] (M2DP), as for
 * pr_sigset_pack.  Stream-ordered (one kernel; PR_HOST buffers are staged and the call then waits).  The raw rows the fp64 re-evaluation reads
 * (pr_rerank_dev's db_sc / db_m2) are the caller's: append them to that buffer too. */
int pr_sigset_reserve(pr_ctx* ctx, pr_sigset* s);
int pr_sigset_append(pr_ctx* ctx, pr_sigset* s, const void* sig, int dtype, int where, int32_t n_new);
/* the packed operand image (DEVICE pointer, read-only), its size and its channel stride in groups / tiles: for tests and for saving a packed DB */
int pr_sigset_image(const pr_sigset* s, const void** image, size_t* bytes, int32_t* channel_stride_groups);

/* processSC.m:22-33 / processM2DP.m:15-21 / processDELIGHT.m:7-37 on packed sets.  d_p, d_i: DEVICE f32 [m][n]
 * (row stride n); DELIGHT writes d_p only (d_i may be NULL). */
int pr_distances_dev(pr_ctx* ctx, const pr_sigset* q, const pr_sigset* db, float* d_p, float* d_i);

/* Per-row moments (one fp64 pass of shifted sums) of a distance shard (first half of MATLAB normalize(.,2), run_test.m:40):
 * mom: DEVICE f64 [m][2][3] = (count, mean, M2 = sum (x-mean)^2) for channel 0 = d_p, 1 = d_i. */
int pr_row_moments_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n, double* mom);

/* run_test.m:38-41 + :47-53 + :57 on a DB shard.  mom_all: DEVICE f64 [G][m][2][3] moments of ALL G shards
 * (combined in rank order, N-1 std); the shard's DB rows are global rows db_row0..db_row0+n-1 and its query
 * rows global q_row0..; mask is |i-j| < mask_width on GLOBAL indices.  idx: DEVICE i32 [m][k] GLOBAL DB
 * indices (-1 = none), score: DEVICE f32 [m][k].  d_i == NULL: plain selection on d_p without fusion (mom_all unused). */
int pr_fuse_select_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n,
                       const double* mom_all, int32_t G, int32_t q_row0, int32_t db_row0, int32_t mask_width,
                       double p_weight, int32_t k, int32_t* idx, float* score);
/* The same, also leaving the (fp32-rounded) scores widened to doubles in score64 DEVICE [m][k] - the form pr_rerank_dev /
 * pr_merge_topk_dev take them in (saves a pr_widen_scores_dev launch; pr_fuse_select2_f64_dev likewise). */
int pr_fuse_select_f64_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n,
                           const double* mom_all, int32_t G, int32_t q_row0, int32_t db_row0, int32_t mask_width,
                           double p_weight, int32_t k, int32_t* idx, float* score, double* score64);
int pr_fuse_select2_f64_dev(pr_ctx* ctx, const float* d_p, const float* d_i, const float* e_p, const float* e_i, int32_t m, int32_t n,
                            const double* mom_all, const double* mom2_all, int32_t G, int32_t q_row0, int32_t db_row0,
                            int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score, double* score64);

/* fp64 re-evaluation of the survivors of pr_fuse_select_dev (run with k_in = k + 8): for every (query, idx_in entry of THIS shard)
 * the distances of the pair again from the RAW signatures in fp64, in the reference's own formulation (processSC.m:15-33: rows / L2
 * norm, 120 shifted / mirrored variants, (1 - dot)/2, min; processM2DP.m:12-22), the fused score of run_test.m:40 with the combined
 * moments, then the k best by (score, index) (run_test.m:57).  q_sc/db_sc: DEVICE [m][2400] / [n_local][2400] or NULL; q_m2/db_m2:
 * DEVICE [4m][384] / [4 n_local][384] or NULL (both pairs given = BASELINE config 5's sum of four z-scores); dtype PR_F64 | PR_F32;
 * mom_sc / mom_m2: DEVICE [G][m][2][3] as pr_row_moments_dev writes them.  idx_in: DEVICE [m][k_in] global indices inside
 * [db_row0, db_row0 + n_local) or -1.  idx: DEVICE [m][k], score: DEVICE f64 [m][k].  k_in <= 128. */
/* Width of the candidate list the top-k protocol re-evaluates for k results in the context's arithmetic: k + 8, or k + 56 in
 * PR_SC_ARITH_F16 (capped at 128). */
int pr_rerank_width(const pr_ctx* ctx, int32_t k);
/* PR_SC_ARITH_F16 only: after pr_rerank_dev / pr_rerank_finish_dev, flags[q] = 1 (DEVICE i32 [m]) for every query whose candidate list
 * (cand_score: the f16 pass scores of the k_in candidates, DEVICE f64 [m][k_in], ascending as pr_fuse_select_dev / pr_merge_topk_dev
 * return them) does not provably contain the exact top-k: exact k-th score (score: DEVICE f64 [m][k]) >= last candidate's pass score
 * minus the score error bound that PR_F16_DISTANCE_BOUND implies with the row's statistics (mom_*: as for pr_rerank_dev).
 * count: DEVICE i32 [1], set to the number of flags.  Flagged queries must be recomputed in PR_SC_ARITH_F16X2.
 * Also flagged: queries whose re-evaluated ORDER is not certain.  A re-evaluated score is exact in the pair's distances, but its
 * channel terms are divided by the f16 pass's row sigmas; two candidates whose channels disagree about their order can change places
 * when those sigmas move by what the pass's distance errors allow (PR_F16_SIGMA_REL + PR_F16_DIST_ERR / sigma).
 * pr_rerank_dev (one shard) / pr_rerank_finish_dev (sharded) check every adjacent pair of the selected k and the best candidate left out and
 * leave the result in the context; this call takes it (once). */
int pr_f16_margin_dev(pr_ctx* ctx, const double* mom_sc, const double* mom_m2, int32_t m, int32_t G, double p_weight, int32_t k_in,
                      const double* cand_score, int32_t k, const double* score, int32_t* flags, int32_t* count);
int pr_rerank_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                  const double* mom_sc, const double* mom_m2, int32_t m, int32_t n_local, int32_t G, int32_t q_row0, int32_t db_row0,
                  int32_t mask_width, double p_weight, int32_t k_in, const int32_t* idx_in, const double* score_in, int32_t k, int32_t* idx,
                  double* score);
/* score_in / cand_score (both forms; DEVICE f64 [m][k_in], may be NULL): the candidates' scores from the fp32 pass, ascending as
 * pr_fuse_select_dev / pr_merge_topk_dev deliver them.  With them a candidate beyond the k-th whose fp32 score exceeds the k-th by more than
 * 64 x the error bound of an fp32 score (from the row statistics) is not re-evaluated - it cannot enter the exact top-k; results are identical.
 *
 * The "p5" block of a re-evaluation, DEVICE f64 [m][5][k_in] per shard: for every query the candidates' scores [k_in], then their four exact
 * channel distances [4][k_in] (SC structure, SC intensity, M2DP count, M2DP intensity; 0 for an absent type; NaN in the first = the pair
 * was not evaluated here: masked (+Inf score), pruned (it keeps its pass score), or another shard's (NaN score)).  pr_rerank_dev keeps its
 * block in the context; the sharded protocol gathers the shards' blocks ([G][m][5][k_in]): it is all the order check and the fp64-statistics
 * resolution need.
 *
 * The sharded form of the re-evaluation (what makes its cost independent of the number of shards): the shards' fp32 top-(k+8) lists
 * are merged FIRST (pr_merge_topk_dev on the gathered lists) into the global candidates cand_idx DEVICE [m][k_in]; every shard then
 * evaluates only the candidates inside its rows [db_row0, db_row0 + n_local) (pr_rerank_partial_dev -> its p5 block), the blocks are
 * all-gathered, and pr_rerank_finish_dev takes each candidate's score from its owner, selects the k best and runs the order and containment
 * checks below (mom_*: the statistics the scores were formed with, [G_mom][m][2][3]; cand_score: the merged pass scores of cand_idx, DEVICE f64
 * [m][k_in] ascending, or NULL = no containment check); the flags stay in the context. */
int pr_rerank_partial_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                          const double* mom_sc, const double* mom_m2, int32_t m, int32_t n_local, int32_t G, int32_t q_row0, int32_t db_row0,
                          int32_t mask_width, double p_weight, int32_t k_in, const int32_t* cand_idx, const double* cand_score, int32_t k,
                          double* p5);
int pr_rerank_finish_dev(pr_ctx* ctx, const double* mom_sc, const double* mom_m2, int32_t G_mom, const int32_t* cand_idx, const double* cand_score,
                         const double* p5_all, int32_t G, int32_t m, int32_t k_in, int32_t k, double p_weight, int32_t* idx, double* score);
/* Two checks, every arithmetic (run_test.m:38-41,57 are fp64 over the WHOLE row; the all-pairs pass is neither).  pr_rerank_dev /
 * pr_rerank_finish_dev leave in the context one word per query:
 *   bit 0  ORDER: two neighbours among the re-evaluated candidates (the selected k and the best one left out) could change places under the
 *          sigma error of the all-pairs pass (their channels disagree about the order and the scores are closer than
 *          sum_c eps_c |z_c(a) - z_c(b)|, eps_c = PR_F32_SIGMA_REL + PR_F32_DIST_ERR / sigma_c; PR_SC_ARITH_F16: the PR_F16_*
 *          constants, and pr_f16_margin_dev takes the flags - such queries go to the split-f16 pass);
 *   bit 1  CONTAINMENT (needs the candidates' pass scores: score_in / cand_score; not in PR_SC_ARITH_F16, whose margin check is
 *          pr_f16_margin_dev): every entry outside the k_in candidates has a pass score >= the last candidate's, T, hence an exact score
 *          >= T - err(T) - (what the sigma error can move it against a listed entry); when the exact k-th best is not below that, an entry
 *          the list does not hold could belong to the top-k - more than k_in entries whose scores agree to the pass's resolution (1e-6 in
 *          the distances): near-copies of one place.
 * A flagged query is answered from its EXACT ROW: its distances to ALL n entries in fp64 -> (count, mean, M2) per channel -> fused scores,
 * mask, the k smallest by (score, index): indices and scores of that query are then those of fp64 arithmetic throughout (run_test.m:38-57),
 * whatever the all-pairs pass made of it.  Three forms, all in passes of 64 flagged queries (ascending):
 *   pr_order_resolve_async_dev  single shard, right after pr_rerank_dev; STREAM-ORDERED, no host synchronisation (fixed-grid kernels that
 *                               leave at once when nothing is flagged; hipGraph-capturable).  ceil(m / 64) passes are chained on the
 *                               stream, so ALL flagged queries are resolved whatever their number (an empty pass costs its launches,
 *                               ~10 us).  mom_sc / mom_m2 rows of resolved queries are overwritten with the exact ones.
 *                               PR_WARN_ORDER_RESOLVED is raised (at pr_take_warnings) when a query was.
 *   pr_order_resolve_dev        the same with a host round trip (reads the count back): ALL flagged queries, *resolved (may be NULL)
 *                               = their number.  The host top-k calls use this one.
 *   sharded                     after pr_rerank_finish_dev, per pass (offset = 0, 64, ...; pr_order_flagged_count gives the total, with a host
 *                               synchronisation; a stream-ordered caller chains ceil(m / 64) passes - empty ones leave at once).  The pass
 *                               called with last_pass != 0 raises PR_WARN_ORDER_UNRESOLVED when flagged queries remain behind it (a
 *                               caller that stops early); no pass ever clears that bit: pr_order_exact_moments_dev = this shard's rows of
 *                               the flagged queries (kept in the context) and their exact (count, mean, M2), exact DEVICE f64 [m][4][3] (rows
 *                               of other queries: unspecified) -> all-gather -> exact_all [G][m][4][3] -> pr_order_exact_select_dev = this
 *                               shard's k best under the statistics of all shards (Chan combination in rank order), sel DEVICE f64 [64][2][k]
 *                               (scores | global indices as doubles, -1 / NaN when the shard has fewer) -> all-gather -> sel_all [G][64][2][k] ->
 *                               pr_order_exact_merge_dev on every rank patches idx / score of the pass's queries (identical inputs, identical
 *                               results everywhere).  G <= 64. */
int pr_order_resolve_async_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                               double* mom_sc, double* mom_m2, int32_t m, int32_t n, int32_t q_row0, int32_t mask_width, double p_weight,
                               int32_t k, int32_t* idx, double* score);
int pr_order_resolve_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                         double* mom_sc, double* mom_m2, int32_t m, int32_t n, int32_t q_row0, int32_t mask_width, double p_weight, int32_t k,
                         int32_t* idx, double* score, int32_t* resolved);
int pr_order_flagged_count(pr_ctx* ctx, int32_t m, int32_t* count);   /* flagged queries of the last m-query pr_rerank_dev / pr_rerank_finish_dev (synchronises) */
int pr_order_exact_moments_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                               const double* mom_sc, const double* mom_m2, int32_t G_mom, int32_t m, int32_t n_local, int32_t offset,
                               int32_t last_pass /* non-zero: the caller runs no pass behind this one */, double* exact);
int pr_order_exact_select_dev(pr_ctx* ctx, const double* exact_all, int32_t G, int32_t m, int32_t n_local, int32_t q_row0, int32_t db_row0,
                              int32_t mask_width, double p_weight, int has_sc, int has_m2, int32_t k, int32_t offset, double* sel);
int pr_order_exact_merge_dev(pr_ctx* ctx, const double* sel_all, int32_t G, int32_t m, int32_t k, int32_t offset, int32_t* idx, double* score);
/* fp32 scores of pr_fuse_select_dev as doubles (the merge works on doubles): DEVICE score32 [count] -> score64 [count] */
int pr_widen_scores_dev(pr_ctx* ctx, const float* score32, int64_t count, double* score64);
/* k-way merge of the per-shard results of G shards (SURVEY.md §8-e collective B's second half): idx_all DEVICE [G][m][k],
 * score_all DEVICE f64 [G][m][k] -> idx [m][k], score [m][k] by (score, global index); -1 / NaN entries last.  PRECONDITION: every shard's list is ASCENDING by (score, index) with its missing entries (-1 / NaN) last - as
 * pr_fuse_select_dev and pr_rerank_dev write them; the merge walks one cursor per list and does not sort (an unsorted list gives a wrong
 * result, not an error).  G <= 64, k <= 128. */
int pr_merge_topk_dev(pr_ctx* ctx, const int32_t* idx_all, const double* score_all, int32_t G, int32_t m, int32_t k,
                      int32_t* idx, double* score);

/* ---- the database row-sharded over several GPUs of one node (the reference is single-device MATLAB,
 * match_signatures/run_test.m:25-57 - this is that computation with hist2 split by rows) ---------------------------------------
 * pr_group_create: one context per entry of device_ids; with G > 1 distinct devices the two exchanges (row moments, per-shard top-k)
 * are ncclAllGather calls of ONE in-process RCCL communicator set (ncclCommInitAll; librccl.so.1 is dlopen'ed here, not linked),
 * enqueued on the compute streams between the kernels.  Device ids may repeat (several shards on one GPU - tests on a one-GPU box):
 * such a group exchanges with event-ordered device copies, since RCCL refuses duplicate devices.  PR_GROUP_EXCHANGE=rccl|copy overrides.
 * pr_group_set_database: hist2 (host f64, [n][2400] SC or [4n][384] M2DP); shard g holds rows [g n/G, (g+1) n/G) raw + packed.
 * pr_group_match_topk: run_test.m:26-57 for hist1 (host f64) against the sharded hist2: idx [m][k] GLOBAL 0-based rows (-1: none),
 * score [m][k] doubles; identical to pr_match_topk_f64 on the unsharded database (same arithmetic, Chan combination in rank order). */
typedef struct pr_group pr_group;
int pr_group_create(const int32_t* device_ids, int32_t G, pr_group** out);
void pr_group_destroy(pr_group* g);
const char* pr_group_last_error(const pr_group* g);     /* g may be NULL (creation errors) */
int32_t pr_group_size(const pr_group* g);
int pr_group_uses_rccl(const pr_group* g);
int32_t pr_group_rccl_ranks(const pr_group* g);         /* ncclCommCount of the group's communicator (0: the group exchanges by copies) */
int pr_group_set_exact_statistics(pr_group* g, int on);  /* pr_set_exact_statistics on every shard: all queries answered from their exact fp64 rows */
int32_t pr_group_last_flagged(const pr_group* g);       /* queries of the last pr_group_match_topk that were answered from their exact fp64 rows */
/* Phase times of a call, per shard (bench.py --via-group: what makes the first multi-GPU curve readable).  on != 0: every following
 * pr_group_match_topk records HIP events on every shard's stream between its phases; pr_group_last_timing waits for the last call's and
 * fills ms [G][PR_GROUP_PHASES] (cap = floats ms has room for), phases in this order:
 *   0 upload + pack(queries) | 1 all-pairs distances | 2 row moments | 3 all-gather A (moments) | 4 fp32 selection | 5 all-gather B (candidate
 *   lists) | 6 merge + fp64 re-evaluation | 7 all-gather C (evaluations) | 8 finish + order / containment checks | 9 exact rows of the flagged
 *   queries (count read back, then per pass: rows, all-gather D, selection, all-gather E, merge).
 * pr_group_create itself runs every exchange size of a one-query call on scratch buffers and checks that every rank's slice arrives in its
 * slot on every device (G > 1): a topology / RCCL problem fails there, with a message, not inside a step. */
#define PR_GROUP_PHASES 10
int pr_group_set_timing(pr_group* g, int on);
int pr_group_last_timing(pr_group* g, float* ms, int32_t cap);
int pr_group_set_database(pr_group* g, int type, const double* h2, int32_t n);
/* A database that grows (SC/test_sc.cpp:40-56 adds one signature row per keyframe; run_test.m:57 matches a frame against all earlier ones):
 * pr_group_set_database_growable = pr_group_set_database with room for extra_capacity more signatures on the LAST shard (whose rows end the
 * global numbering; its operand image is laid out for that capacity, pr_sigset_reserve); pr_group_append_database adds hist_new ([n_new][2400]
 * SC or [4 n_new][384] M2DP, host f64) there in place - no re-pack, no re-upload of what is resident - and the next pr_group_match_topk
 * sees n + n_new rows.  With one device this is the online loop of a single GPU through host buffers only. */
int pr_group_set_database_growable(pr_group* g, int type, const double* h2, int32_t n, int32_t extra_capacity);
int pr_group_append_database(pr_group* g, const double* hist_new, int32_t n_new);
int32_t pr_group_database_rows(const pr_group* g);
int pr_group_take_warnings(pr_group* g);                /* OR of the shards' pr_take_warnings (PR_WARN_* bits), then cleared */
int pr_group_match_topk(pr_group* g, const double* h1, int32_t m, int32_t mask_width, double p_weight, int32_t k, int32_t* idx,
                        double* score);

/* Device-buffer variants of the generators (same layouts as the host versions, pointers in HBM). */
int pr_sc_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                       double max_rho, double* out);
int pr_m2dp_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                         double max_rho, double* out);
int pr_delight_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* out);
/* The generators with the clouds' PCA frames (utils/pts_align.h:7-46) supplied by the caller: the moments pass over the points is skipped
 * and only the binning pass runs (28 B per point once instead of twice).  frames: device [N][16] doubles = mean[3], the three
 * eigenvectors by ascending eigenvalue [9], 0, point count, then optionally the cloud's float intensity average (the reference's
 * sequential float sum in input order, SC.cpp:60-64 / M2DP.cpp:77-81, widened to double) and 1.0 where it is there.
 * pr_cloud_frames_dev writes them (asynchronously, on the context's streams; with inten != NULL the averages too), and
 * pr_pts_preprocess_gpu leaves them - averages included - beside the clouds it emits (pr_clouds_dev_frames).
 * frames_have_ave != 0: slots 14 of the frames hold the averages and the call is the binning pass alone; 0: the call computes them
 * (a chain of dependent float adds per cloud, beside the binning pass).  Same results, bit for bit, as the calls above. */
int pr_cloud_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten /* may be NULL */, const int64_t* offs, int32_t N, double* frames);
int pr_sc_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho,
                              const double* frames, int frames_have_ave, double* out);
int pr_m2dp_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho,
                                const double* frames, int frames_have_ave, double* out);
int pr_delight_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                                   const double* frames, double* out);

/* ---- host-side rows a1/a2 (CPU in the reference too; no device, no context) ---------------------------- */

/* Replaces pts_preprocess(poses_file, pts_file, incoming_id_file, lidarRange, clouds, polar_filter)
 * (utils/pts_preprocess.h:169-232, records PosesPts.h:5-40): sliding world-point window per pose, camera-frame
 * transform, range cut, voxel (polar_filter = 0, SC) or 1-degree polar (polar_filter = 1, M2DP) down-sampling with the
 * reference's point ORDER, and the incoming_id_file (written when non-NULL).  verbose prints the reference's lines. */
int pr_pts_preprocess(const char* poses_file, const char* pts_file, const char* incoming_id_file, double lidarRange,
                      int polar_filter, int verbose, pr_clouds** out);
/* The same on the GPU: identical clouds in identical point order (the order of libstdc++'s
 * std::unordered_map iteration, reproduced from the key insertion sequence), files parsed on the host.  Needs a context. */
int pr_pts_preprocess_gpu(pr_ctx* ctx, const char* poses_file, const char* pts_file, const char* incoming_id_file,
                          double lidarRange, int polar, int verbose, pr_clouds** out);
/* Iteration order of a std::unordered_map<int,...> after inserting the K distinct non-negative keys in this order
 * (order[t] = index of the t-th element); the host build of the routine the GPU pre-stage runs per cloud. */
int pr_hash_order(const int32_t* keys, int32_t K, int32_t* order);
int64_t pr_clouds_count(const pr_clouds* c);
const int64_t* pr_clouds_offs(const pr_clouds* c);      /* [N+1] */
const double* pr_clouds_xyz(const pr_clouds* c);        /* [offs[N]][3] camera frame */
const float* pr_clouds_inten(const pr_clouds* c);       /* [offs[N]] */
const int32_t* pr_clouds_ids(const pr_clouds* c);       /* [N] incoming ids */
double pr_clouds_avg_ms(const pr_clouds* c);           /* the reference's "average time" of pts_preprocess.h:221-225 (files already parsed) */
double pr_clouds_avg_pts(const pr_clouds* c);
/* Clouds made by pr_pts_preprocess_gpu also stay in HBM (device of that context) with their PCA frames, laid out as the host arrays
 * above, for pr_*_generate_frames_dev; NULL for clouds made on the host (and for an empty result).  Freed by pr_clouds_free. */
const double* pr_clouds_dev_xyz(const pr_clouds* c);
const float* pr_clouds_dev_inten(const pr_clouds* c);
const int64_t* pr_clouds_dev_offs(const pr_clouds* c);
const double* pr_clouds_dev_frames(const pr_clouds* c);  /* [N][16] */
/* The loop of SC/test_sc.cpp:40-56 / M2DP/test_m2dp.cpp:41-68 / DELIGHT/test_delight.cpp:41-56 over a pr_clouds object: type = PR_TYPE_*,
 * out = host [N][2400] | [4N][384] | [16N][256] (max_rho is ignored for DELIGHT).  Clouds that pr_pts_preprocess_gpu left in this
 * context's HBM are taken from there with their frames (no upload, binning pass only); any other pr_clouds goes the way of
 * pr_sc_generate / pr_m2dp_generate / pr_delight_generate from its host arrays.  Same signatures either way. */
int pr_generate_clouds(pr_ctx* ctx, int type, const pr_clouds* c, double max_rho, double* out);
void pr_clouds_free(pr_clouds* c);

/* Signature matrix text I/O: writer = `ofstream << Eigen::MatrixXd` (test_sc.cpp:63-66, test_m2dp.cpp:83-86);
 * reader = whitespace-tolerant load (test_kitti.m:26); *out is released with pr_free. */
int pr_write_signatures(const char* path, const double* sig, int64_t rows, int64_t cols);
int pr_read_signatures(const char* path, double** out, int64_t* rows, int64_t* cols);
void pr_free(void* p);
/* Binary side-car of the same matrices (no reference counterpart): 32-byte header + row-major data,
 * dtype PR_F64 or PR_F32 on disk; the reader always returns f64.  The executables pick it by the ".bin" suffix. */
int pr_write_signatures_bin(const char* path, const double* sig, int64_t rows, int64_t cols, int dtype);
int pr_read_signatures_bin(const char* path, double** out, int64_t* rows, int64_t* cols);
/* PosesPts.h:12-24 / :36-39 record writers (the producer side, OutputWrapperSODSO.cpp:24-31). */
int pr_write_poses(const char* path, const int32_t* ids, const double* w2c, int64_t n);
int pr_write_points(const char* path, const int32_t* ids, const double* xyz, const float* inten, int64_t n);
/* Replaces the evaluation half of run_test(type, hist1, hist2, gt1, gt2, loop_diff, mask_width) (match_signatures/run_test.m:3-22 ground-truth
 * loop pairs, :58-85 precision / recall sweep): diff_v / diff_idx [m] = the per-query best score and 0-based index (run_test.m:57),
 * gt1 [m][cols], gt2 [n][cols] positions.  auc = trapz(recall, precision), top_recall = recall at the last 100 %-precision point,
 * lp_detected (optional) [m][2] receives the *n_detected pairs (query, match) of that prefix.  A diff_idx of -1 (no finite candidate) is
 * read as index 0: MATLAB's min over an all-NaN / all-Inf row returns index 1 (run_test.m:57). */
int pr_precision_recall(const double* diff_v, const int32_t* diff_idx, int32_t m, const double* gt1, const double* gt2, int32_t n,
                        int32_t cols, double loop_diff, int32_t mask_width, double* auc, double* top_recall, int32_t* lp_detected,
                        int32_t* n_detected);
const char* pr_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PLACE_RECOGNITION_H */
