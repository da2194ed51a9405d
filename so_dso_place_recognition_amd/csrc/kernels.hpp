// kernels.hpp — launch wrappers of the gfx950 kernels (internal to libpr_amd.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pr {

// ------------------------------------------------------------------ packed layouts
constexpr int SC_NF = 31;                       // rfft bins of the 60 sectors
constexpr int SC_NSLOT = 16;                    // frequency pairs per DB group: (0,30),(1,2),...,(27,28),(29,-)
constexpr int SC_QIMG = SC_NF * 320;            // floats per (channel, 8-query group): [pos(31)][s(5)][lane(64)]
constexpr int SC_DSTEP = 640;                   // floats per (channel, 16-entry DB group, frequency position)
constexpr int SC_DIMG = 2 * SC_NSLOT * SC_DSTEP;  // floats per (channel, 16-entry DB group): 32 positions, the last one zero
// frequency -> position in the packed images (processing order 0,30,1,2,...,29)
__host__ __device__ inline int sc_fpos(int f) { return f == 0 ? 0 : (f == 30 ? 1 : f + 1); }
// split-f16 images (sc_match_h.hip), sizes in BYTES
constexpr int SCH_QBLK = 1288;                  // (8-query group, frequency): 16 rows x 80 B (Q hi | Q lo), rows 8..15 shifted by 8 B
constexpr int SCH_QIMG = SC_NF * SCH_QBLK;      // 39 928 per (channel, 8-query group)
constexpr int SCH_DTILE = 768;                  // one 16x16x32 column operand: 48 lanes x 16 B
constexpr int SCH_DFREQ = 4 * SCH_DTILE;        // Re hi, Re lo, Im hi, Im lo
constexpr int SCH_DIMG = SC_NF * SCH_DFREQ;     // 95 232 per (channel, 16-entry DB group)
// The three split products q_hi d_hi + q_hi d_lo + q_lo d_hi of a frequency's 20 rings are 60 terms; two 16x16x32 MFMAs have 64 K-slots, so
// the matchers run them as TWO operand pairs (not three zero-padded ones) whose 16-byte lane pieces are all contiguous in these images:
//   query row (80 B)   Q hi ring 0..19 | Q lo ring 16..19, 0..15      (the lo half rotated by four rings: Q hi 16..19 and Q lo 16..19 adjoin)
//   DB hi tile         lanes 0-15 D hi 0..7 | 16-31 D hi 8..15 | 32-47 D hi 16..19, zero
//   DB lo tile         lanes 0-15 D lo 0..7 | 16-31 D lo 8..15 | 32-47 D lo 16..19, D hi 16..19   (a copy of the hi values where zeros stood)
//   pair 1:  A lanes (k = lane >> 4) at row + {0, 16, 32, 0}    x  B = hi tile lanes 0-47, lo tile lanes 0-15
//            = Qhi Dhi (rings 0..19) + Qhi Dlo (0..7)                                        [k = 2: Q lo 16..19 meets the tile's zeros]
//   pair 2:  A lanes at row + {48, 64, 32, 16}                 x  B = hi tile lanes 0-31, lo tile lanes 32-47, lo tile lanes 16-31
//            = Qlo Dhi (0..15) + Qhi Dlo (16..19) + Qlo Dhi (16..19) + Qhi Dlo (8..15)
// The single-product kernels read the hi tiles' 48 lanes and the first 40 bytes of a query row - unchanged by this.
__host__ __device__ constexpr int sch_qlo_byte(int ring) { return 40 + ((ring + 4) % 20) * 2; }      // Q lo of a ring, from the row's first byte
__host__ __device__ constexpr int sch_a1_byte(int k) { return k == 3 ? 0 : 16 * k; }                   // pair 1 / pair 2: this lane group's 16 B of the row
__host__ __device__ constexpr int sch_a2_byte(int k) { return k == 0 ? 48 : k == 1 ? 64 : k == 2 ? 32 : 16; }
__host__ __device__ constexpr int sch_b1_byte(int lane) { return lane < 48 ? lane * 16 : SCH_DTILE + (lane - 48) * 16; }   // from the hi tile's first byte
__host__ __device__ constexpr int sch_b2_byte(int lane) { return lane < 32 ? lane * 16 : lane < 48 ? SCH_DTILE + lane * 16 : SCH_DTILE + (lane - 32) * 16; }
// single-product f16 images (PR_SC_ARITH_F16, sc_match_e.hip): the hi halves only
constexpr int SCF_QBLK = 640;                   // (8-query group, frequency): 16 rows x 40 B, NO shift of rows 8..15: the rows are read with ds_read2_b64
                                                // (banks = dword mod 32, 16 consecutive lanes per access) and 10 r mod 32 is distinct for r = 0..15 - the 8-byte shift
                                                // the 80-byte rows of the split image need made rows 8, 9, 10 collide with rows 5, 6, 7 here (round 6: SQ_LDS_BANK_CONFLICT)
constexpr int SCF_QIMG = SC_NF * SCF_QBLK;      // 20 088 per (channel, 8-query group); a workgroup holds 8 groups = 64 queries
constexpr int SCF_DFREQ = 2 * SCH_DTILE;        // Re, Im
constexpr int SCF_DIMG = SC_NF * SCF_DFREQ;     // 47 616 per (channel, 16-entry DB group) = 2976 B per entry and channel
constexpr int M2_TILE = 96 * 64;                // floats per (channel, 32-row tile): [kq(24)][lane(64)][4]

inline int sc_qgroups8(int m) { return ((m + 31) / 32) * 4; }
inline int sc_qgroups8_f16(int m) { return ((m + 63) / 64) * 8; }
inline int sc_dgroups(int n) { return (n + 15) / 16; }
inline int m2_tiles(int sigs) { return (sigs + 7) / 8; }      // 8 signatures x 4 variants = 32 rows
inline int m2_qtiles(int sigs) { return ((m2_tiles(sigs) + 11) / 12) * 12; }   // query tiles: workgroups take 3 or 4 of them

// Binary intensity channel (SC.cpp:67-72 writes 0 / 1 there): when every row of channel 1 has all of its non-zero entries equal (and
// positive), the row normalised as processSC.m:15-20 does is 1/sqrt(ones) on `ones` bins and every one of the 120 variant products of
// processSC.m:30 is count / sqrt(ones_q ones_d) with an INTEGER count.  The split-f16 pack then also leaves
//   binfo[row] = {sqrt(ones), 1/sqrt(ones)} (floats; {1, 1} for a zero-norm row)
//   bstat[0] |= 1 when a row of channel 1 is not of that form, bstat[1] = max ones, bstat[2..5] = max over the rows of the
//   w-weighted square sum of the f16 rounding residuals of the row's hi spectra (float bits; one slot per frequency block of the pack kernel:
//   their sum bounds the largest row's residual norm^2)
// (bstat is zeroed by the caller in front of every pack) and the matcher runs channel 1 with ONE f16 product per term on the hi halves
// of the same images and rounds count = max x sqrt(ones_q ones_d) to the nearest integer - exact for every pair that passes the test of
// ep_store_round (sc_match_e.hip) under the error bound sc_bin_bound() evaluates from these numbers; the split-f16 kernel otherwise.
constexpr int SC_BSTAT_INTS = 8;
struct ScBin {                    // what the matcher needs of the two sets (device pointers), by value in the kernel arguments
  const int* qstat; const int* dstat;
  const float* qinfo; const float* dinfo;     // [rows][2]
  int* viol;                      // [1] set to `gen` by the single-product pass when a pair fails its rounding test (ep_store_round)
  int gen;                        // the call's sequence number (> 0): the split-f16 pass behind looks for exactly this value, so nobody has to clear the word
  float bconst;                   // (u + gamma)(1 + u) + slack: S and stage-2 constant rounding (pr_api.cpp: create_common)
  float pair_scale;               // 1; tests: PR_SC_BINARY_PAIR_SCALE inflates the bound of the per-pair test only, so that the pass runs and fails it
  int gate;                       // 0: always run; 1: the single-product pass (runs when the bound predicts success); 2: the split-f16 pass behind it (runs when that one did not, or raised viol)
  int chsel;                      // -1: both channels (channel = XCD & 1); 0 / 1: this channel on all XCDs
};
#ifdef __HIPCC__
// The error bound of the single-product pass over a binary channel, evaluated by every workgroup from the same numbers (derivation in
// DESIGN.md §4.0b): with eq, ed the largest w-weighted residual norms of the hi spectra of the two sets, every variant product of a pair is
// within sqrt(ones_q ones_d) x bound of its integer count, bound = eq + ed + eq ed + bconst (1 + eq)(1 + ed).  Returns whether the pass is
// worth running: both sets binary and the largest pair's bound below 0.72 of a count (the pass's actual error, ~6 x 7.6e-5 per unit of
// sqrt(ones_q ones_d) at most, then leaves the per-pair test of ep_store_round room).
__device__ __forceinline__ bool sc_bin_bound(const ScBin& b, float& bound) {
  const int q0 = b.qstat[0], q1 = b.qstat[1], d0 = b.dstat[0], d1 = b.dstat[1];
  const float eq = __builtin_sqrtf(((__int_as_float(b.qstat[2]) + __int_as_float(b.qstat[3])) + __int_as_float(b.qstat[4])) + __int_as_float(b.qstat[5]));
  const float ed = __builtin_sqrtf(((__int_as_float(b.dstat[2]) + __int_as_float(b.dstat[3])) + __int_as_float(b.dstat[4])) + __int_as_float(b.dstat[5]));
  bound = (eq + ed + eq * ed + b.bconst * (1.f + eq) * (1.f + ed)) * 1.001f;
  return q0 == 0 && d0 == 0 && bound * __builtin_sqrtf((float)q1 * (float)d1) < 0.72f;
}
#endif
// sc_pack.hip — processSC.m:15-20 (row L2 normalisation) + per-ring rfft over the 60 sectors, written in the
// MFMA operand layout of `role`.  sig: device [rows][2400] of T.  flags[0] |= 1 if a row has zero norm.
// bad[row] |= 1 << channel for such rows (they are packed as zeros; launch_nan_fixup writes their NaN distances).
void launch_sc_pack(hipStream_t st, const void* sig, int dtype, int rows, int role, float* packed, int groups,
                    const double* twiddle, int* flags, int* bad);
// sc_match.hip — processSC.m:22-33 for both channels.  Writes d_p, d_i device [m][n].
void launch_sc_match(hipStream_t st, const float* qpk, int m, const float* dpk, int n, const float* cst,
                     float* d_p, float* d_i, int nsplit_override);
size_t sc_match_lds_bytes();
void launch_zero_ints(hipStream_t st, int* p, int n);
void launch_fill_ints(hipStream_t st, int* p, int n, int v);
// sc_match_h.hip — the same on the f16 matrix cores with split (hi + lo) operands; packed images from launch_sc_pack_h
void launch_sc_pack_h(hipStream_t st, const void* sig, int dtype, int rows, int role, void* packed, int groups,
                      const double* twiddle, int* flags, int* bad, int single = 0,   // single: hi halves only (SCF_* layout)
                      float* binfo = nullptr, int* bstat = nullptr);                 // binary-channel statistics (ScBin above)
void launch_sc_pack_h_rows(hipStream_t st, const void* sig, int dtype, int rows, int row0, void* packed, int groups, const double* twiddle,
                           int* flags, int* bad, int single, float* binfo, int* bstat);   // rows [row0, row0 + rows) of a DB image (pr_sigset_append)
void launch_sc_match_h(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p,
                       float* d_i, int nsplit_override, const struct ScBin* bin = nullptr);   // bin: channel selection / gate of a binary-channel call
size_t sc_match_h_lds_bytes();
// sc_match_e.hip — the pair-walk form (no row-exchanged query operand; see the file): single = 0: split-f16 (three products, 4 waves),
// single = 1: one f16 product per term (PR_SC_ARITH_F16), 8 waves = two per SIMD; same packed images and constants as sc_match_h.hip
void launch_sc_match_e(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override, int single, int dgs = 0 /* channel stride of the DB image in 16-entry groups when it is not sc_dgroups(n): an appendable set */);
// split-f16 images whose channel 1 may be binary (ScBin above): channel 0 in split-f16; channel 1 by the single-product kernel on the hi
// halves with integer rounding when the sets' statistics allow it, in split-f16 otherwise - the decision is taken ON THE DEVICE by every
// workgroup from the same numbers (no host round trip: the two channel-1 launches are both issued, one of them leaves at once).
// ev (or null): four events recorded around the launches (channel 0 | channel 1 single product | channel 1 split)
// (Measured and dropped: the channel-0 launch of an online call and the binary pass on two streams at once - 0.302 -> 0.308 ms at m = 1,
// 0.310 -> 0.337 at m = 8: sc_match_h's workgroups fill the LDS of their CUs, the two kernels do not run side by side, and the event hand-offs cost.)
void launch_sc_match_e_bin(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                           int nsplit_override, ScBin bin, hipEvent_t* ev, int online_h = 0 /* m <= 8: sc_match_h.hip's one-group form (PR_SC_ONLINE=h, read once at pr_create) */, int dgs = 0 /* as in launch_sc_match_e */);

// m2dp_match.hip — processM2DP.m:12-22 for both channels.
void launch_m2dp_pack(hipStream_t st, const void* sig, int dtype, int sigs, float* packed, int tiles);
void launch_m2dp_match(hipStream_t st, const float* qpk, int m, const float* dpk, int n, float* d_p, float* d_i);
// m2dp_match_h.hip — the same with split-f16 operands on the f16 matrix cores (tiles of the same size, packed by launch_m2dp_pack_h)
void launch_m2dp_pack_h(hipStream_t st, const void* sig, int dtype, int sigs, void* packed, int tiles, int sg0 = 0 /* first signature's place in the image */);
void launch_m2dp_match_h(hipStream_t st, const void* qpk, int m, const void* dpk, int n, float* d_p, float* d_i, int single = 0 /* hi halves only */,
                         int dts = 0 /* channel stride of the DB image in tiles when it is not m2_tiles(n): an appendable set */);

// fuse_select.hip — run_test.m:38-41,47-53,57
// scratch (select_scratch_bytes(), or null): lets a call with few query rows cut every row into slices (grid m x P) instead of one workgroup per row
size_t select_scratch_bytes();
void launch_row_moments(hipStream_t st, const float* d_p, const float* d_i, int m, int n, double* mom, void* scratch = nullptr);
void launch_fuse_select(hipStream_t st, const float* d_p, const float* d_i, int m, int n, const double* mom_all,
                        int G, int q_row0, int db_row0, int mask_width, double p_weight, int k, int32_t* idx,
                        float* score, const float* e_p = nullptr, const float* e_i = nullptr, const double* mom2_all = nullptr, void* scratch = nullptr,
                        double* score64 = nullptr /* the same scores widened to f64 [m][k] */);

// rerank.hip — NaN rows / columns of zero-norm signatures (processSC.m:16,19), the fp64 re-evaluation of the fp32
// selection's survivors (processSC.m:15-33 / processM2DP.m:12-22 + run_test.m:40 per pair) and the k-way shard merge
void launch_nan_fixup(hipStream_t st, float* d_p, float* d_i, int m, int n, const int* qbad, const int* dbad);
// p5 [m][5][kin] (p5_all [G][m][5][kin]): per query the candidates' scores [kin] and their four exact channel distances [4][kin] (SC structure,
// SC intensity, M2DP count, M2DP intensity; NaN in the first = not evaluated) - what a shard knows after its re-evaluation (rerank.hip)
void launch_rerank(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                   const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                   double p_weight, int kin, const int32_t* idx_in, double* p5, unsigned* tick /* [cap] zeros (left zero) | [cap] work list | [1] */, size_t tick_cap /* >= m kin */, int k, int32_t* idx, double* score,
                   float* score32, const double* cand_sc32, double eps_d = 0.0, double order_floor = 0.0,
                   double order_noise = 0.0, int32_t* order_flags = nullptr);   // order_flags [m]: the order check of launch_order_check on the result; cand_sc32: the candidates' all-pairs-pass scores (ascending) or null: prunes hopeless candidates; eps_d: that pass's distance error bound (0: the fp32-grade 1e-6 with a 64x margin)
// the sharded form: scores + distances of the candidates THIS shard owns (NaN elsewhere), then owner-wise combination + selection
void launch_rerank_partial(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                           const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                           double p_weight, int kin, const int32_t* idx_in, double* p5, unsigned* tick, size_t tick_cap, const double* cand_sc32, int k,
                           double eps_d = 0.0);
// PR_SC_ARITH_F16: flags[q] = 1 where the candidate list does not provably contain the exact top-k (rerank.hip), count += number of flags
void launch_margin_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int G, int m, double p_weight, int kin,
                         const double* cand_sc, int k, const double* score, double eps_d, int32_t* flags, int32_t* count,
                         const int32_t* order_flags, double eps_floor, double noise /* the order check's constants of this arithmetic: sigma slack of the containment test */);
// flags[q] = 1 where the order of the re-evaluated candidates (the selected k and the best one left out) could change under the sigma error
// of the all-pairs pass (per channel eps_floor + noise / sigma: noise = the largest error of one distance; statistics from mom_* [Gmom][m][2][3])
// + bit 1 where the candidate list (cand_sc [m][kin]: its all-pairs-pass scores, ascending; score_sel [m][k]: the exact scores of the k selected;
// both or neither) does not provably hold the exact top-k of the whole row
void launch_order_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int Gmom, const int32_t* cand_idx, const double* p5_all,
                        int G, int m, int kin, int k, const int32_t* idx_sel, double p_weight, double eps_floor, double noise, int32_t* flags,
                        const double* cand_sc = nullptr, const double* score_sel = nullptr);
// The flagged queries (order not certain under the pass's sigmas | candidate list not provably complete: rerank.hip) answered from their EXACT
// rows, stream-ordered (exact_row.hip): flags -> ascending list + count; per pass of RESOLVE_SLOTS list slots from `offset`
//   launch_xrow         this shard's fp64 distances of the slot's query to every entry (rows [RESOLVE_SLOTS][4][n_local]) and their exact
//                       (count, mean, M2) per channel (exact [m][4][3]; partial: scratch [RESOLVE_SLOTS][RESOLVE_NB][4][3])
//   launch_xrow_select  the k smallest (fused fp64 score, global index) of the slot's row under the statistics of all shards (exact_all
//                       [G][m][4][3]) and the mask -> sel [RESOLVE_SLOTS][2][k] (scores | indices as doubles); with sel = null (one shard)
//                       straight into idx / score [m][k], out_mom_* rows [m][2][3] patched with the exact moments
//   launch_xrow_merge   sel_all [G][RESOLVE_SLOTS][2][k] of all shards -> idx / score of the slots' queries
constexpr int RESOLVE_SLOTS = 64;
constexpr int RESOLVE_NB = 2048;
constexpr int RESOLVE_SMALL_M = 1024;            // calls of up to this many queries: the kernels scan the flags themselves (no compaction launch)
int exact_partial_blocks(int n_local);
void launch_flag_compact(hipStream_t st, const int32_t* flags, int m, int32_t* list, int32_t* cnt);
// compacted: list / cnt are already there (launch_flag_compact: the host-synchronising form runs several passes over one list)
void launch_xrow(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                 const double* mom_sc, const double* mom_m2, int G, int m, int n_local, const int32_t* flags, int32_t* list, int32_t* cnt,
                 int offset, bool compacted, double* partial, double* exact, double* rows, unsigned* tick, int* dflags,
                 int last_pass /* no pass follows: flagged queries beyond this pass's 64 are reported in dflags[3] */, double* qspec /* [xrow_qspec_doubles()] scratch: the slots' query spectra */, const double* tw /* cos | sin of 2 pi t / 60 */,
                 int direct /* 1: the reference's own formulation instead of the spectral form */);
size_t xrow_qspec_doubles();
hipError_t xrow_set_twiddles(const double* cos60, const double* sin60);   // once per device, before the first launch_xrow (constant memory of exact_row.hip)
void launch_xrow_select(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* exact_all, int G,
                        int m, int n_local, int q_row0, int db_row0, int mask_width, double p_weight, int has_sc, int has_m2, int k,
                        const double* rows, double* part /* [xrow_select_part_doubles(m, k)] scratch: the row slices' lists */, double* sel, int32_t* idx,
                        double* score, double* out_mom_sc, double* out_mom_m2);
size_t xrow_select_part_doubles(int m, int k);
void launch_xrow_merge(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* sel_all, int G, int m,
                       int k, int32_t* idx, double* score);
void launch_rerank_finish(hipStream_t st, const int32_t* cand_idx, const double* p5_all, int G, int m, int kin, int k, int32_t* idx,
                          double* score);
void launch_widen(hipStream_t st, const float* a, long long n, double* b);
void launch_merge_topk(hipStream_t st, const int32_t* idx_all, const double* score_all, int G, int m, int k, int32_t* idx, double* score);

// sc_gen.hip / m2dp_gen.hip — pts_align.h:7-46 + SC.cpp:12-76 / M2DP.cpp:38-109 (+ test_m2dp.cpp:44-68)
void launch_ave_chain(hipStream_t st, const float* inten, const int64_t* offs, int N, float* ave /* [N] or NULL */,
                      double* frames = nullptr /* non-NULL: also frames[c][14] = the average, [15] = 1 (frames.hpp) */);
void launch_cloud_frames(hipStream_t st, const double* xyz, const int64_t* offs, int N, double* frames);
// the one-HBM-pass path of SC generation (see sc_gen.hip): a batch of clouds small enough for the Infinity Cache, W workgroups per cloud
constexpr int SC_MAX_W = 16;                     // slices per cloud
constexpr int SC_MAX_SPLIT_CLOUDS = 1024;        // clouds per batch when W > 1
constexpr size_t SC_SCRATCH_PER_STREAM = (size_t)2 * SC_MAX_SPLIT_CLOUDS * 4 + (size_t)SC_MAX_SPLIT_CLOUDS * SC_MAX_W * 9 * 8 +
                                         (size_t)768 * (1200 * 4 + 3 * 1200 * 8);   // tickets + partial moments + partial grids (nb * W <= 768)
size_t sc_generate_scratch_bytes();
void launch_sc_batch(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int c0, int c1, int W, double max_rho,
                     double* frames, char* scratch, double* out);
size_t sc_cluster_scratch_bytes(int ncl, int CW);
int sc_cluster_points_per_workgroup();
int* launch_sc_cluster(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N, double max_rho, int CW, int ncu,
                       char* scratch, double* frames, double* out);
void launch_sc_finish(hipStream_t st, const float* ave, int N, double* out);
void launch_sc_bin(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N, double max_rho,
                   const double* frames, const float* ave /* NULL: out[c][1200..] = bin means, launch_sc_finish applies the averages */, double* out,
                   int ave_in_frames = 0 /* with ave = NULL: the averages are frames[c][14] */);
void launch_m2dp_bin_svd(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                         double max_rho, const double* frames, const float* ave, const double* planes, double* mats,
                         double* out, int* flags /* [1] |= 1: a leading singular pair did not converge */,
                         int* svd_rows /* [0] += 1 and [1 + slot] = signature row (cloud * 4 + variant) for each such pair */,
                         hipStream_t st2 = nullptr, hipEvent_t* ev_bin = nullptr, hipEvent_t* ev_svd = nullptr /* [2] each: more than one
                         batch of clouds -> the singular pairs of a batch run on st2 beside the binning of the next (see m2dp_gen.hip) */);
constexpr int M2DP_SVD_ROWS_CAP = 1024;
size_t m2dp_generate_scratch_bytes(int N);

// plain_match.hip — processGIST.m:1-10, processBoW.m:1-38 (h1, h2: device, row-major doubles; BoW rows alternate ids | weights)
void launch_gist_distance(hipStream_t st, const double* h1, int m, const double* h2, int n, int cols, float* dist);
void launch_bow_distance(hipStream_t st, const double* h1, int m, const double* h2, int n, int cols, float* dist);
// delight.hip — DELIGHT.cpp:8-24, processDELIGHT.m:1-38
void launch_delight_gen(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                        const double* frames, double* out);
void launch_delight_pack(hipStream_t st, const void* sig, int dtype, int sigs, float* packed, unsigned* mask);
void launch_delight_match(hipStream_t st, const float* q, int m, const float* db, const unsigned* dbmask, int n, float* dist);

// prestage.hip — utils/pts_preprocess.h:135-232 on the GPU (see the file header); all pointers are device pointers
int64_t prestage_cells(double range, int polar);          // dense cell-table length per pose
void launch_death(hipStream_t st, const double* xyz, const int* birth, int64_t T, const double* W, const unsigned char* emit,
                  const int* next_reset, double range, int* death);
void launch_members(hipStream_t st, int E, const int* death, const int* pose_of, const int64_t* first_alive, const int64_t* cursor,
                    const int64_t* off, int* cnt, int* list);
void launch_cells(hipStream_t st, const double* xyz, const int* list, const int64_t* off, const int* pose_of, int e0, int e1,
                  int64_t s0, int64_t s1, const double* W, double range, int polar, int64_t C, int* cell, unsigned long long* val,
                  unsigned long long* tval, unsigned* tfirst, unsigned* tbest);
void launch_keys(hipStream_t st, const int64_t* off, int e0, int e1, int64_t C, const int* cell, const int* list, const unsigned* tfirst,
                 const unsigned* tbest, int* keys, int* win, int* nkeys);
void launch_order(hipStream_t st, int E, const int64_t* off, const int* nkeys, const int* keys, const int* sched_cnt, const int* sched_nb,
                  int nsched, int* next, const int64_t* boff, int* bkt, int* order);
void launch_gather(hipStream_t st, int E, int64_t total, const int64_t* off, const int64_t* ooff, const int* pose_of, const int* order,
                   const int* win, const double* xyz, const float* inten, const double* W, double range, double* oxyz, float* oint,
                   double* frames /* non-NULL: one workgroup per cloud, which also leaves the cloud's PCA frame [E][16] (frames.hpp) */);

}  // namespace pr
