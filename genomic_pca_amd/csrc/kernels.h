// Launchers for the gfx950 kernels (kernels.hip).  Every launcher only enqueues work on `st`;
// none allocates or synchronises.  Shapes are validated by the callers in gpca_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpca {

// Row pitch of the int8 genotype matrix and row count of Q/Y buffers: multiple of this many samples.
constexpr int64_t kSamplePad = 256;   // = samples covered by one wave tile of the G^T T kernel
constexpr int kGQRowsPerWave = 128;   // SNP rows per wave in the G Q kernel (R = 4 tiles of 32)

struct QcParams { double min_call_rate, min_maf, max_hwe_p; };

// Kernel-selection switches of one handle (read from the environment at gpca_create; defaults = the measured best).
struct KernelOpts {
    int gq_phase = -1;   // workgroup b of k_gq_d starts its sweep of the sample axis at stage ((b % 8) A + (b / 8) B) mod stages, A = low 16 bits,
                         // B = high 16 bits; -1 = the launcher's choice (8 starting points per XCD up to 51k samples).  Harness knob.
    int gq_chain = 1;    // k_gq_d prefetches a wave's next round behind the current round's epilogue (0: drain + prologue per round).  Harness knob.
    int gtt_xcd = 1;     // XCD-aware workgroup order in k_gtt_d / k_gtt_p.  Harness knob.
};
// Opt-in to > 64 KiB of dynamic LDS for every kernel that needs it, on the CURRENT device (the attribute is per device).
// Return a hipError_t value (0 = ok).
int init_device_kernels_i8();
int init_device_kernels_common();
int init_device_kernels_eig();

// ---- the small dense step on the device (small_eig.hip) ----------------------------------------------------------------------
constexpr int kMaxSketchCols = 128;                  // = kMaxSketch (gpca_internal.h)
// layout of the result block `res` of launch_small_eigh (doubles): what the host reads back ONCE, at the end of the call
constexpr int kEigResSv = 0;                         // [128] singular values sqrt(max(w, 0)), descending
constexpr int kEigResEig = kMaxSketchCols;           // [128] w_c / denom for c < k
constexpr int kEigResW = 2 * kMaxSketchCols;         // [128] eigenvalues w, descending
constexpr int kEigResFlag = 3 * kMaxSketchCols;      // [0] the CholeskyQR pivot flag, [1] QL sweep cap hit
constexpr int kEigResCount = 3 * kMaxSketchCols + 8;
// Symmetric eigenproblem of the leading n x n block of W (pitch L; nslices > 0: W = the sum of `nslices` partial matrices [L * L] apart),
// symmetrised; descending.  Z [2][L][k]: zmode 0 -> Z0 = V_k diag(sv), Z1 = V_k diag(1 / sv); zmode 1 -> Z0 = Z1 = V_k.  Vout may be NULL.
void launch_small_eigh(hipStream_t st, const double* src, int nslices, int n, int L, int k, int zmode, double denom, const int* cholflag,
                       double* Z, double* res, double* Vout);

// Blocked layouts of the skinny GEMM operands (gemm_f32.hip): a lane's 8 / 16 k-steps are contiguous.
//   Tb [group = row/16][lt][lane = 32*(row&1) + col%32][u = (row%16)/2]      (8 floats per lane)
//   Qb [chunk = n/32][lt][lane = 32*((n%32)/16) + col%32][u = n%16]           (16 floats per lane)
__host__ __device__ inline int64_t blocked_t_index(int64_t row, int col, int LT) {
    const int64_t group = row >> 4;
    const int w = (int)(row & 15), lt = col >> 5, cc = col & 31;
    return ((group * LT + lt) * 64 + ((w & 1) * 32 + cc)) * 8 + (w >> 1);
}
__host__ __device__ inline int64_t blocked_q_index(int64_t n, int col, int LT) {
    const int64_t chunk = n >> 5;
    const int w = (int)(n & 31), lt = col >> 5, cc = col & 31;
    return ((chunk * LT + lt) * 64 + ((w >> 4) * 32 + cc)) * 16 + (w & 15);
}

void launch_synth(hipStream_t st, int8_t* G, int64_t M, int64_t N, int64_t ld, int64_t snp_offset,
                  uint64_t seed, const uint32_t* d_thresh, int P);
// Fast panel generator (GPCA_PANEL_SYNTH16): one 16-bit uniform per genotype (SplitMix64, counter mode), g = (u < t1) + (u < t2) with the 16-bit
// thresholds t2 = P(g = 2) (low half) and t1 = P(g >= 1) (high half) of thresh16[row][(n / 16) % P]; 4 genotypes per SplitMix64 output.
// packed = 0: int8 rows of pitch ld; 1: 2-bit dosage codes, rows of pitch ld bytes.  snp0 = global index of row 0.
void launch_synth16(hipStream_t st, void* G, int packed, int64_t rows, int64_t N, int64_t ld, int64_t snp0, uint64_t seed,
                    const uint32_t* d_thresh16, int P);
void launch_bed_decode(hipStream_t st, const uint8_t* bed, int64_t bytes_per_row, int8_t* G, int64_t M,
                       int64_t N, int64_t ld);
// a1: per-SNP {n_valid,n0,n1,n2}, mu, sigma, r = 1/sigma, b = -mu r, keep, reason; flags[0] |= 1 if any
// kept SNP holds a missing value, |= 2 if any kept SNP holds a value outside {0,1,2}.
void launch_snp_stats(hipStream_t st, const int8_t* G, int64_t M, int64_t N, int64_t ld, QcParams qc,
                      float* mu, float* sigma, float* r, float* b, uint8_t* keep, uint8_t* reason,
                      uint32_t* counts, uint32_t* flags);
// recompute r, b, flags from caller-supplied mu/sigma/keep (needs counts from a stats pass for the flags)
void launch_set_scale(hipStream_t st, int64_t M, const float* mu, const float* sigma, const uint8_t* keep,
                      float* r, float* b);
// a2: out[a][c] = fma((f32)g, 1/sigma, -mu/sigma); err_idx = min flat index of a missing genotype (or -1)
void launch_standardize_block(hipStream_t st, const int8_t* G, int64_t ld, const float* mu,
                              const float* sigma, const int64_t* rows, int64_t ns, const int64_t* cols,
                              int64_t nj, float* out, unsigned long long* err_idx);

// sketch operand: Tb (blocked, all Mpad rows) = r_i * Omega[i][j] (j < l, else 0); cpart[wave][j] = sum_i b_i Omega[i][j]
int64_t omega_num_parts(int64_t Mpad);
// row_ids (may be NULL): global index of every row (a matrix of gathered rows draws the normals its rows would draw in place)
void launch_omega(hipStream_t st, int64_t M, int64_t Mpad, int l, int L, int64_t snp_offset, uint64_t seed,
                  const float* r, const float* b, float* Tb, float* cpart, double* apart, int blocked = 1, const int64_t* row_ids = nullptr);
void launch_gather_rows(hipStream_t st, const void* src, int64_t pitch, const int64_t* ids, int64_t n, void* dst);      // dst row i <- src row ids[i]
void launch_gather_elems(hipStream_t st, const void* src, int elem_bytes, const int64_t* ids, int64_t n, void* dst);   // 1-, 4- or 16-byte elements

// exact-integer path: the digit planes of T' = r o Omega directly (analytic column bound 6.67 * rmax[0]; no f32 T')
void launch_omega_planes(hipStream_t st, int64_t M, int64_t Mpad, int l, int L, int64_t snp_offset, uint64_t seed, const float* r,
                         const float* b, float* cpart, int8_t* Td, const float* rmax, double* tscale, double* tinv, int nd,
                         const int64_t* row_ids = nullptr);
// out[0] = max_i r[i] for r >= 0 (the caller zeroes out first)
void launch_max_f32(hipStream_t st, const float* r, int64_t n, float* out);

// K1: T = r o (G Q) + b s^T.  Tb != NULL: write r o T blocked into Tb and cpart[wave][j] = sum_i b_i T_ij
// (power iteration); Tb == NULL: write T row-major into Tout (projection B = A Q).  Qb is the blocked basis.
struct GqPlan { int64_t units; int64_t waves; };   // 32-row units of the padded matrix; resident waves (multiple of 4)
GqPlan gq_plan(int64_t Mpad, int waves_target);
// G: int8 rows (packed = 0) or 2-bit dosage codes (packed = 1) of row pitch ldr bytes; Npad = padded sample count (multiple of 256)
void launch_gq_f32(hipStream_t st, const void* G, int packed, int64_t ldr, const GqPlan& plan, int64_t Npad, const float* Qb,
                   int L, const float* r, const float* b, const float* s, float* Tout, float* Tb, float* cpart);
// K2: Ypart[w][n][j] = 2^-9 * sum over the wave's SNP rows of G[i][n] * T'[i][j]   (T' blocked in Tb)
struct GttPlan { int64_t nblocks_n; int W; int64_t rows_per_wave; int64_t grid; };
GttPlan gtt_plan(int64_t Mpad, int64_t Npad, int L, int target_waves);
void launch_gtt_f32(hipStream_t st, const void* G, int packed, int64_t ldr, int64_t Mpad, int64_t Npad, const float* Tb,
                    int L, float* Ypart, const GttPlan& plan);
// Y[n][j] = c[j] + 2^9 * sum_w Ypart[w][n][j]   (f64 sum), n < N
void launch_reduce_y(hipStream_t st, const float* Ypart, int W, int64_t Npad, int64_t N, int L, const double* c,
                     double* Y);

// generic tall-skinny helpers
// scratch: >= S * E doubles, S <= 64 slices (<= 256 when E <= 64): kSumScratchElems covers E <= 4096
constexpr int64_t kSumScratchElems = 64 * 4096;
void launch_sum_partials_f32(hipStream_t st, const float* part, int64_t P, int64_t E, double* out, double* scratch);
void launch_sum_partials_f64(hipStream_t st, const double* part, int64_t P, int64_t E, double* out, double* scratch);
// slices of the part axis the first stage uses (a function of (P, E) only: the summation tree is the same for every run and partition)
int sum_slices(int64_t P, int64_t E);
#ifdef __HIPCC__
// The body of the two-stage sum (one workgroup of 256 threads = columns 64 bx .. + 63, slice by of S): out[by][e] = sum over the slice's
// parts in a fixed tree.  Shared by k_sum_partials and by the kernels that carry one of its stages along (k_post_k1, k_quantize).
template <typename T>
__device__ __forceinline__ void sum_partials_body(const T* __restrict__ part, int64_t P, int64_t E, double* __restrict__ out, int S, int bx, int by,
                                                  double (*red)[64]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t e = (int64_t)bx * 64 + lane;
    const int64_t per = (P + S - 1) / S;
    const int64_t p0 = by * per, p1 = (p0 + per < P) ? p0 + per : P;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (e < E) {
        int64_t p = p0 + wv;
        for (; p + 12 < p1; p += 16) {
            a0 += (double)part[p * E + e]; a1 += (double)part[(p + 4) * E + e];
            a2 += (double)part[(p + 8) * E + e]; a3 += (double)part[(p + 12) * E + e];
        }
        for (; p < p1; p += 4) a0 += (double)part[p * E + e];
    }
    red[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wv == 0 && e < E) out[(int64_t)by * E + e] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}
#endif
// first stage only: *slices partial sums of [E] doubles at *src, for a consumer that folds the (<= 64, or <= 256 for E <= 64) slices
// itself (launch_small_eigh, k_chol_inv); P <= 64 parts are their own slices and nothing is launched
void launch_sum_partials_f64_stage1(hipStream_t st, const double* part, int64_t P, int64_t E, double* scratch, const double** src, int* slices);
int64_t gram_num_parts(int64_t rows);
void launch_gram_f64(hipStream_t st, const double* X, int64_t rows, int L, double* part);
void launch_gram_f32(hipStream_t st, const float* X, int64_t rows, int L, double* part);
// X[n][:] <- X[n][:] * Z  (Z: [L][L] f64, in place); optionally also the blocked f32 basis Qb (rows_pad rows, pad rows zeroed)
// the digit planes hold round(x * S / colmax): S = 0.49 * 128^4 keeps every digit of the signed base-128 expansion in int8
constexpr double kDigitScale = 0.49 * 268435456.0;
// three-plane mode (packed kernels only): round(x * S3 / colmax) in three signed base-256 digits, S3 = 0.49 * 256^3
constexpr double kDigitScale3 = 0.49 * 16777216.0;
inline double digit_scale(int nd) { return nd == 3 ? kDigitScale3 : kDigitScale; }
// last right-multiplication of CholeskyQR2 + partials of Q^T 1 and of the column abs-max; k_finish_q reduces them
int64_t tail_num_parts(int64_t rows_pad);
void launch_apply_right_tail(hipStream_t st, double* X, int64_t rows, int L, const double* Z, float* Qout, int64_t rows_pad,
                             double* csum_part, double* amax_part);
void launch_finish_q(hipStream_t st, const double* csum_part, const double* amax_part, int64_t P, int L, double* s64, float* s32,
                     double* scale, double* inv, int nd = 4);
// W = R^T R (n x n, pitch ld <= 64), Z = R^-1; *flag = j + 1 on a non-positive pivot (first failure wins)
void launch_chol_inv(hipStream_t st, const double* W, int n, int ld, double* Z, int* flag);
// the same from the Gram's P <= 64 partial sums [P][32 * 32] (ld = 32 only): the fold rides in front of the factorisation, no k_sum_partials launch
void launch_chol_inv_fold(hipStream_t st, const double* part, int P, int n, int ld, double* Z, int* flag);
void launch_apply_right_inplace(hipStream_t st, double* X, int64_t rows, int L, const double* Z, float* Qb,
                                int64_t rows_pad);
// out[n][kc] = sum_j X[n][j] Z[j][kc]   (Z: [L][K] f64)
void launch_rightmul_f64(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64,
                         float* out32);
// rows gathered through an index list (loadings of kept SNPs); sign (may be NULL): column c of Z is multiplied by sign[c]
void launch_rightmul_gather_f32(hipStream_t st, const float* X, const int64_t* row_ids, int64_t nrows, int L,
                                const double* Z, int K, float* out32, const int* sign = nullptr);
// Sample scores with their sign rule in two launches (L <= 64): launch_scores writes the unsigned scores X Z and, per workgroup and
// column, the first row with maximal |score| (value + row: scores_num_parts(rows) candidates); launch_scores_sign folds the candidates
// (every workgroup for itself: they are few), flips the columns whose winner is negative (in place, plus the f32 copy) and leaves
// sign[c] = +-1 for the loadings.
int64_t scores_num_parts(int64_t rows);
void launch_scores(hipStream_t st, const double* X, int64_t rows, int L, const double* Z, int K, double* out64, double* cand_val, int64_t* cand_idx);
void launch_scores_sign(hipStream_t st, double* out64, float* out32, int64_t rows, int K, const double* cand_val, const int64_t* cand_idx,
                        int64_t parts, int* sign);
int64_t colsum_num_parts(int64_t rows);
void launch_colsum_f64(hipStream_t st, const double* X, int64_t rows, int L, double* part);
// sign[c] = sign of the first element of column c with maximal |x|
void launch_col_sign(hipStream_t st, const double* X, int64_t rows, int K, int* sign);
void launch_scale_cols(hipStream_t st, double* X64, float* X32, int64_t rows, int K, const int* sign);
void launch_f64_to_f32(hipStream_t st, const double* in, float* out, int64_t n);
// exclusive scan of keep -> row list of kept SNPs (small M only needs one block; large M uses 2 passes)
void launch_fill_f32(hipStream_t st, float* p, int64_t n, float v);
// Tp[rows[a]][c] = load[a][c] for c < k (other entries untouched: caller zero-fills Tp first)
void launch_expand_loadings(hipStream_t st, const float* load, const int64_t* rows, int64_t n_pca, int k, int L, float* Tp);
// Tb (blocked, all Mpad rows) = r_i X[i][:]; cpart[wave][j] = sum over the wave's 64 rows of b_i X[i][j] (omega_num_parts(Mpad) waves)
void launch_scale_rows(hipStream_t st, const float* X, int64_t M, int64_t Mpad, int L, const float* r, const float* b,
                       float* Tb, float* cpart, int blocked = 1);


// EigenSNP stages: block-diagonal condensed basis W (kernels.hip)
void launch_bd_expand(hipStream_t st, const float* W, const int32_t* feat0, int cmax, const double* P, int64_t M, int L, float* out);
void launch_bd_reduce(hipStream_t st, const float* W, const int32_t* feat0, int cmax, const float* T, int L, const int64_t* blk_row0,
                      const int64_t* blk_row1, const int32_t* blk_feat0, const int32_t* blk_c, int B, double* P);
void launch_rightmul_inplace_f32(hipStream_t st, float* X, int64_t rows, int L, const double* Z);
void launch_mask_rows(hipStream_t st, double* Y, int64_t N, int L, const uint8_t* mask);
void launch_f32_to_f64(hipStream_t st, const float* in, double* out, int64_t n);

// ---- exact-integer path (gemm_i8.hip), 32 columns per launch.  Every K1 here writes cpart[unit][32] = the unit's share of
// b^T T (one partial per 32-row unit of the launch, partition-independent) and, except k_gq_i8, apart[wave][32] = column abs-max. ----------------------------------------------------------
constexpr int kDigits = 4;   // signed base-128 digits of the skinny operand
void launch_gq_i8(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t N, const int8_t* Qd,
                  const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart,
                  int scale_out, int64_t ldt = 32, const KernelOpts& ko = KernelOpts());
// K2 work decomposition: tasks = (row chunk, n-group) pairs -- an n-group is the 4 x 128 samples of a workgroup's four waves, a row chunk
// one of W near-equal ranges of the 128-row stages -- and every n-group ends with W partial tiles in Ypart[W][Npad][32].
// Simple kernels / narrow shapes: one task per workgroup (grid = tasks).  The DMA kernels (k_gtt_d, k_gtt_p) take `tasks_per_wg`
// CONSECUTIVE tasks per workgroup, n-group fastest, so that a launch is one batch of workgroups that end together: one task per
// workgroup ran 500 workgroups on 256 CUs at configs[1] and 392 at configs[3]'s shard -- a second, part-empty batch, 6 % and 14 % of
// the launch.  gtt8_plan_batched picks W (stages per task against per-task prologues, the fold's traffic and the L2 footprint of a
// row chunk's T' planes, which the workgroups of an XCD share).
struct Gtt8Plan { int64_t nblocks_n; int W; int64_t rows_per_wave; int64_t grid; int tasks_per_wg; int64_t S; int64_t ngroups; int strided; int64_t C; };
Gtt8Plan gtt8_plan(int64_t Mpad, int64_t Npad, int target_waves);
Gtt8Plan gtt8_plan_batched(int64_t Mpad, int64_t Npad, int target_waves);
void launch_gtt_i8(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td,
                   double* Ypart, const Gtt8Plan& plan, const KernelOpts& ko = KernelOpts());
// narrow matrices (N <= 256 samples, int8 rows): every wave owns its own row range, Q's digit planes stay in registers (K1) /
// the four waves of a workgroup take four row chunks instead of four sample blocks (K2).  launch_gq_n returns a hipError_t value.
constexpr int64_t kNarrowSamples = 256;
Gtt8Plan gtt8_plan_narrow(int64_t Mpad, int64_t N, int target_waves);
void launch_gtt_n(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Ypart,
                  const Gtt8Plan& plan);
int launch_gq_n(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t N, const int8_t* Qd,
                const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart, double* apart,
                int scale_out, int64_t ldt = 32);
void launch_reduce_y_i8(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, const double* c,
                        const double* tscale, double* Y, int64_t ldy = 32);
void launch_accum_y_i8(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, double* Yint, int first);
void launch_finish_y_i8(hipStream_t st, const double* Yint, int64_t N, const double* c, const double* tscale, double* Y, int64_t ldy = 32);
void launch_absmax_fold(hipStream_t st, const double* apart, int64_t P, double* run);
void launch_accum_y_scaled(hipStream_t st, const double* Ypart, int W, int64_t Npad, int64_t N, const double* tscale, double* Yacc, int first);
void launch_finish_y_sum(hipStream_t st, const double* Yacc, int64_t N, const double* c, double* Y, int64_t ldy = 32);
int64_t absmax_num_parts(int64_t rows);
// X [rows][32] row-major -> digit planes Xd [rows_pad/32][kDigits][64][16 B]; scale[j] = colmax_j / S, inv = 1/scale
// layout 0: 32 consecutive rows per block; layout 1: the MFMA-step order of the packed (2-bit) G Q kernel
void launch_quantize_f32(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, double* part, double* scale,
                         double* inv, int8_t* Xd, int layout = 0, int nd = 4, int64_t ldx = 32);
void launch_quantize_f64(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, double* part, double* scale,
                         double* inv, int8_t* Xd, int layout = 0, int nd = 4, int64_t ldx = 32);

// ---- 2-bit resident genotypes (store2bit.hip, gemm_i8.hip) -----------------------------------------------------
constexpr int64_t kSamplePad2bit = 1024;   // samples per row padded to this (ld2 = Npad / 4 bytes)
void launch_bed_to_codes(hipStream_t st, const uint8_t* bed, int64_t bpr, uint8_t* G2, int64_t M, int64_t N, int64_t ld2);
void launch_pack_i8(hipStream_t st, const int8_t* G8, int64_t ld8, uint8_t* G2, int64_t rows, int64_t N, int64_t ld2,
                    unsigned* flags);
void launch_snp_stats_2bit(hipStream_t st, const uint8_t* G2, int64_t M, int64_t N, int64_t ld2, QcParams qc, float* mu,
                           float* sigma, float* r, float* b, uint8_t* keep, uint8_t* reason, uint32_t* counts, uint32_t* flags);
void launch_standardize_block_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, const float* mu, const float* sigma,
                                   const int64_t* rows, int64_t ns, const int64_t* cols, int64_t nj, float* out,
                                   unsigned long long* err_idx);
void launch_gq_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, const GqPlan& plan, int64_t Npad, const int8_t* Qd,
                    const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart,
                    double* apart, int scale_out, int nd = 4, int64_t ldt = 32);
void launch_quantize_f64_prescaled(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, const double* inv, int8_t* Xd, int layout, int nd = 4, int64_t ldx = 32);
// the same with launch_finish_q folded in for these 32 columns (every workgroup folds the P abs-max partials itself: for P <= kFinishQFoldMax)
constexpr int64_t kFinishQFoldMax = 512;
void launch_quantize_f64_finishq(hipStream_t st, const double* X, int64_t rows, int64_t rows_pad, int8_t* Xd, int layout, int nd, int64_t ldx,
                                 const double* csum_part, const double* amax_part, int64_t P, int ldp, double* s64, float* s32, double* scale, double* inv);
// quantise X whose column abs-max partials [P][32] were already produced by the kernel that wrote it (K1 epilogue)
void launch_quantize_f32_premax(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, const double* apart, int64_t P,
                                double* scale, double* inv, int8_t* Xd, int layout, int nd = 4, int64_t ldx = 32);
// What follows a K1 sweep of a power iteration, in two launches instead of four (resident matrices):
//   launch_post_k1:           first stage of c = b^T T over the units' partials cpart [units][32] (S = post_k1_slices(units) slices into
//                             cscratch) BESIDE the fold of the waves' column abs-max apart [P][32] into the digit scale (one more workgroup);
//   launch_quantize_f32_cfold: the digit planes of T' as launch_quantize_f32_premax's second half, and -- its first workgroup -- the
//                             second stage of c (cscratch -> c[32]).  Same summation tree as launch_sum_partials_f32: the same bits.
int post_k1_slices(int64_t units);
void launch_post_k1(hipStream_t st, const float* cpart, int64_t units, double* cscratch, const double* apart, int64_t P, double* scale, double* inv, int nd);
void launch_quantize_f32_cfold(hipStream_t st, const float* X, int64_t rows, int64_t rows_pad, const double* inv, int8_t* Xd, int layout, int nd,
                               int64_t ldx, const double* cscratch, int cslices, double* c);
// K2 for packed genotypes: stage-wise cooperative LDS-DMA (ring of four stage buffers); returns a hipError_t value
int launch_gtt_p(hipStream_t st, const uint8_t* G2, int64_t ld2, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Ypart,
                 const Gtt8Plan& plan, int nd = 4, const KernelOpts& ko = KernelOpts());
// K2 with genotypes and digit planes brought in by LDS-DMA (int8-resident); returns a hipError_t value (0 = ok)
int launch_gtt_d(hipStream_t st, const int8_t* G, int64_t ldg, int64_t Mpad, int64_t Npad, const int8_t* Td, double* Ypart,
                 const Gtt8Plan& plan, const KernelOpts& ko = KernelOpts());
// K1 with the genotypes brought in by LDS-DMA (full-line pieces); returns a hipError_t value (0 = ok)
int launch_gq_d(hipStream_t st, const int8_t* G, int64_t ldg, const GqPlan& plan, int64_t Npad, const int8_t* Qd,
                const double* qscale, const float* r, const float* b, const float* s, float* Tout, float* cpart, double* apart,
                int scale_out, int64_t ldt = 32, const KernelOpts& ko = KernelOpts());
void launch_gtt_2bit(hipStream_t st, const uint8_t* G2, int64_t ld2, int64_t Mpad, int64_t Npad, const int8_t* Td,
                     double* Ypart, const Gtt8Plan& plan, int nd = 4);

}  // namespace gpca
