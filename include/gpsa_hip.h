/* gpsa_hip.h — C ABI of libgpsa_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * GPSA variational deep-GP forward / ELBO / backward hot path.
 *
 * Boundary contract
 *   - plain C: device pointers + sizes + a hipStream_t passed as void*; no torch types.
 *   - every entry point returns 0 on success or a (positive) hipError_t / negative GPSA_E* code;
 *     nothing allocates, frees or synchronises (safe for stream capture into a hipGraph);
 *     scratch memory is passed in by the caller (size from the matching *_workspace() query).
 *   - all matrices are dense row-major.  dtype: GPSA_F32 (float) or GPSA_F64 (double).
 *   - scalar kernel hyper-parameters are passed as DEVICE pointers (unconstrained = log values,
 *     as the reference stores them) so that no host<->device sync is ever needed.
 *
 * Each entry point cites the reference code (python, /root/reference) it replaces.  The reference
 * has no FFI of its own (pure PyTorch); the binding a maintainer would add is the ctypes stub shown
 * in INTEGRATION.md and implemented in spatial_alignment_amd/_lib.py.
 */
#ifndef GPSA_HIP_H
#define GPSA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  GPSA_F32 = 0,
  GPSA_F64 = 1,
  GPSA_F32_X64 = 2,   /* gpsa_kmat in_dtype only: fp32 Z / hyper-parameters, fp64 X */
  GPSA_F32_OUT64 = 3, /* gpsa_kmat_bwd in_dtype only: fp32 inputs, fp64 gradients (with dtype = GPSA_F64) */
  GPSA_F32_ACC64 = 4  /* gpsa_kmat_bwd in_dtype only: fp32 inputs AND an fp32 Kbar, fp64 arithmetic / partial
                         sums / gradients (with dtype = GPSA_F64) */
};
enum { GPSA_K_RBF = 0, GPSA_K_MATERN12 = 1, GPSA_K_MATERN32 = 2 };
enum { GPSA_EINVAL = -1, GPSA_EWORKSPACE = -2, GPSA_EUNSUPPORTED = -3 };

/* library / build info */
int gpsa_version(void);               /* 100*major + minor */
const char* gpsa_build_arch(void);    /* "gfx950" */
const char* gpsa_source_hash(void);   /* "GPSA_SOURCE_HASH=<sha256 of csrc/ + this header at build time>": the host
                                         side refuses a library that was not built from the sources next to it */

/* ---- covariance ("kernel") matrices ---------------------------------------------------------
 * K[m,c] = k(Z[m,:], X[c,:]) (+ jitter on the diagonal m==c, used for K_uu).
 * replaces gpsa/util/util.py:8-23 (rbf_kernel), :33-47 (matern12_kernel), :50-66 (matern32_kernel)
 * as called from gpsa/models/vgpsa.py:314-318, 390-392, 409.
 * Z [M,D], X [C,D], K [M,C]; ls_u / var_u: device scalars (log lengthscale, log variance). D <= 4.
 * dtype = type K is computed and stored in; in_dtype = storage type of Z, X, ls_u, var_u (the fp32
 * parameters are read as they are).  Supported (dtype, in_dtype): (F32,F32), (F64,F64), (F64,F32) and
 * (F64, F32_X64): fp32 Z / hyper-parameters with an fp64 X - the data GP's k(Gtilde, G_samples) on the warp
 * GP's unrounded draws. */
int gpsa_kmat(int dtype, int in_dtype, int kind, const void* Z, int M, const void* X, long long C,
              int D, const void* ls_u, const void* var_u, double jitter, void* K, void* stream);

/* Backward of gpsa_kmat: given Kbar = dLoss/dK [M,C] (dtype) produce, in in_dtype,
 *   dZ [M,D], dX [C,D] (may be NULL), dparams[2] = {dLoss/d ls_u, dLoss/d var_u}.
 * same != 0: Z and X are the same points (K_uu, C == M): the X-side sums are added into dZ and dX is
 * not written.  (autograd of util.py:8-66 in the reference).  Deterministic: per-workgroup partial sums
 * (in dtype) in the workspace, then ONE second launch that adds them in a fixed order.
 * in_dtype = GPSA_F32_OUT64 (dtype GPSA_F64): fp32 inputs, dZ / dX / dparams stored as fp64 - the data GP's
 * backward, whose sigma^2 and coordinate gradients are differences of large terms; GPSA_F32_ACC64: the
 * same with Kbar itself stored as fp32 (the data GP's gradient panel). */
long long gpsa_kmat_bwd_workspace(int dtype, int M, long long C, int D);
int gpsa_kmat_bwd(int dtype, int in_dtype, int kind, const void* Z, int M, const void* X, long long C,
                  int D, const void* ls_u, const void* var_u, const void* Kbar, int same, void* dZ,
                  void* dX, void* dparams, void* workspace, long long workspace_bytes, void* stream);

/* ---- dense products ---------------------------------------------------------------------------
 * C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b];  op(A): m x k, op(B): k x n, row-major, strided batch.
 * splitk > 1 splits the k loop over workgroups (deterministic: partials in workspace, then summed).
 * replaces the torch.matmul / torch.mm calls of vgpsa.py:179-196, 207-210, 227, 302, 430. */
long long gpsa_gemm_workspace(int dtype, int m, int n, int batch, int splitk);
int gpsa_gemm(int dtype, int transA, int transB, int m, int n, long long k, double alpha,
              const void* A, long long lda, long long strideA, const void* B, long long ldb,
              long long strideB, double beta, void* C, long long ldc, long long strideC, int batch,
              int splitk, void* workspace, long long workspace_bytes, void* stream);

/* out[M,C] += A[M,L] B[L,C] (fp32, row-major, contiguous): a thin inner dimension under a long panel, in ONE pass
 * over the panel - the mean term's share of the data GP's projection gradient, autograd of vgpsa.py:192
 * (abar += delta_F dmean^T: M = 200, L = 50, C = 100k at the headline size).  GPSA_EUNSUPPORTED outside M <= 256,
 * L <= 64, C >= 4096 a multiple of 4, out 16-byte aligned: use gpsa_gemm with beta = 1. */
int gpsa_thin_update_f32(const float* A, int M, int L, const float* B, long long C, float* out, void* stream);

/* ---- variational covariances (vgpsa.py:206-210): Omega[b] = A[b] A[b]^T + jitter I ---------------
 * A [batch,M,M] is the fp32 parameter (Omega_sqt_*), read as stored; Omega [batch,M,M] fp64 (matrix
 * cores).  gpsa_omega_bwd is its adjoint: dA[b] = (G[b] + G[b]^T) A[b] with G = dLoss/dOmega (fp64);
 * symmetric != 0 promises G = G^T and computes 2 G A (no strided reads of G^T). */
int gpsa_omega_fwd(const float* A, int M, int batch, double jitter, double* Omega, void* stream);
int gpsa_omega_bwd(const double* G, const float* A, int M, int batch, int symmetric, float* dA,
                   void* stream);
/* The same over TWO parameter tensors in one launch (the warp GPs' Omega_sqt_G_list, vgpsa.py:131-143, and a
 * modality's Omega_sqt_F_dict entry, :145-153, which share M when m_X_per_view == m_G): segment 0 has n0
 * matrices, segment 1 n1 (0: absent). */
int gpsa_omega_fwd2(const float* A0, int n0, double* Omega0, const float* A1, int n1, double* Omega1, int M,
                    double jitter, void* stream);
int gpsa_omega_bwd2(const double* G0, const float* A0, float* dA0, int n0, const double* G1, const float* A1,
                    float* dA1, int n1, int M, int symmetric, void* stream);

/* ---- inducing-point factorisations (fp64, batched, one workgroup per matrix) -----------------
 * gpsa_chol_f64: in-place lower Cholesky of A[b] (upper triangle zeroed); logdet[b] = 2*sum(log diag);
 *   info[b] = 0 or (1 + index of the first non-positive pivot) — the reference raises
 *   torch.linalg.LinAlgError there (vgpsa.py:257, 320, 394, 412 torch.cholesky).
 * gpsa_tri_inv_f64: Linv[b] = inverse of the lower-triangular L[b] (replaces the triangular solves
 *   inside torch.cholesky_solve, vgpsa.py:177). */
int gpsa_chol_f64(void* A, int M, int batch, void* logdet, int* info, void* stream);
int gpsa_tri_inv_f64(const void* L, void* Linv, int M, int batch, void* stream);
/* gpsa_chol_inv_f64: Linv[b] = chol(A[b])^-1 with logdet / info as gpsa_chol_f64, in one register-
 *   resident sweep (A is not modified; M <= 256, GPSA_EUNSUPPORTED above: chain the two calls above).
 *   Same reference lines as the pair it fuses. */
int gpsa_chol_inv_f64(const void* A, void* Linv, int M, int batch, void* logdet, int* info,
                      void* stream);
/* gpsa_chol_inv_sel_f64: the same for a SELECTION of a batch in one launch (M <= 256): the matrices
 *   b < n_always and keep_lo <= b < keep_hi of A [batch,M,M]; results land at their own index b in Linv,
 *   logdet and info, the other entries are not touched.  A data-parallel rank that owns only a range of the
 *   KL terms of vgpsa.py:498-530 factorises the priors and ITS variational covariances (owner computes; the
 *   gradient all-reduce sums the shares) - still one launch, i.e. one matrix's latency. */
int gpsa_chol_inv_sel_f64(const void* A, void* Linv, int M, int batch, int n_always, int keep_lo, int keep_hi,
                          void* logdet, int* info, void* stream);
/* gpsa_chol_inv_blocked_f64: the same result for any M: right-looking over diagonal blocks of <= 256
 *   columns (register-resident kernel per block, fp64-MFMA products for the panel, the trailing update and
 *   the rows of the inverse).  workspace >= gpsa_chol_inv_blocked_workspace(M, batch) bytes. */
long long gpsa_chol_inv_blocked_workspace(int M, int batch);
int gpsa_chol_inv_blocked_f64(const void* A, void* Linv, int M, int batch, void* logdet, int* info,
                              void* workspace, long long workspace_bytes, void* stream);

/* ---- EXPERIMENT (not on the product path, never the timed step; csrc/split_bf16.hip) ------------------------------
 * The contraction W_l = Omega_l alpha of vgpsa.py:192-196 with every fp32 operand written as a sum of bf16 pieces and
 * the product as bf16 matrix instructions accumulated in fp32 (fp32 MFMA is 1/16 of the bf16 rate on gfx950):
 *   nprod 6: three pieces, the six products a_i b_j with i + j <= 4;  4: two pieces, four products;  3: two pieces
 *   without a2 b2;  1: plain bf16 operands;  0: the fp32 instruction on the same tiling (comparison).
 * Omega [L,M,M], alpha [M,C], W [L,M,C] fp32.  A plain kernel (operands from global memory): numerics, not speed.
 * gpsa_experiment_split_bf16_rate: the MFMA + LDS-fragment-read loop of the fused ELBO kernel's tile (13 x 2
 * accumulators, ``outputs`` outputs per workgroup, 256 workgroups) in fp32 (nprod 0) or split form (6 / 4 / 3); the
 * caller times it; out >= 65536 floats. */
int gpsa_experiment_split_bf16_product(const float* Omega, const float* alpha, int M, long long C, int L, int nprod,
                                       float* W, void* stream);
int gpsa_experiment_split_bf16_rate(int nprod, int outputs, float* out, void* stream);

/* ---- the dominant contraction: variational variance term --------------------------------------
 * v[l,c] = alpha[:,c]^T Omega[l] alpha[:,c]            (vgpsa.py:192-196 a_t_Omega_tril, square, sum;
 *                                                        Omega_tril Omega_tril^T == Omega exactly)
 * alpha [M,C], Omega [L,M,M] symmetric, v [L,C].  Never materialises the [S,L,N,M] tensor.
 * fp32 + M <= 256 runs on the MFMA (v_mfma_f32_16x16x4_f32) path; otherwise a tiled generic path.
 * omega_dtype: storage type of Omega (the fp64 Omega = A A^T + 1e-5 I is read as stored and rounded
 * to the compute type while it is packed for the matrix cores). */
long long gpsa_quadform_workspace(int dtype, int M, long long C, int L);
int gpsa_quadform_fwd(int dtype, int omega_dtype, const void* alpha, const void* Omega, int M,
                      long long C, int L, void* v, void* workspace, long long workspace_bytes,
                      void* stream);
/* dalpha[:,c] = 2 * sum_l g[l,c] * Omega[l] alpha[:,c]      (autograd of the above wrt alpha) */
int gpsa_quadform_bwd_alpha(int dtype, int omega_dtype, const void* alpha, const void* Omega,
                            const void* g, int M, long long C, int L, void* dalpha, void* workspace,
                            long long workspace_bytes, void* stream);
/* The same form for a layer with few outputs (the warp GP: L = D <= 3), keeping the products
 * W[l] = Omega[l] alpha ([L,M,C], dtype) it is made of, so that its backward is one streaming pass
 *   dalpha[:,c] = 2 * sum_l g[l,c] * W[l][:,c]
 * instead of L more M x M x C products.  Omega in dtype.  (vgpsa.py:192-196 and its autograd.)
 * dcT [M,L] / meanT [L,C] (both or neither NULL): the layer's mean term meanT[l,c] = sum_m dcT[m,l] alpha[m,c]
 * (vgpsa.py:182-184) in the same pass over alpha that closes the form. */
int gpsa_quadform_fwd_keep(int dtype, const void* alpha, const void* Omega, int M, long long C, int L,
                           void* v, void* W, const void* dcT, void* meanT, void* stream);
/* dcT [M,L] / dmeanT [L,C] (both or neither NULL): adds the mean term's share dcT dmeanT to dalpha. */
int gpsa_quadform_bwd_alpha_kept(int dtype, const void* W, const void* g, int M, long long C, int L,
                                 const void* dcT, const void* dmeanT, void* dalpha, void* stream);
/* The data GP's form (many outputs, fp32 matrix cores, M <= 256) with its products kept: v as gpsa_quadform_fwd
 * and, in the same pass, the products Omega[l] alpha themselves into W - an opaque buffer of
 * gpsa_quadform_keep_f32_bytes(M, C, L) bytes (about L M C 4) in the kernel's own accumulator order, so that each
 * wave stores 1 KiB contiguous.  Training's backward is then gpsa_quadform_bwd_alpha_kept_f32:
 *     dalpha[:,c] = 2 sum_l g[l,c] (Omega[l] alpha)[:,c]  ( + dcT dmeanT: the mean term's share, dcT [M,L] and
 *     dmeanT [L,C] both or neither NULL, as in gpsa_quadform_bwd_alpha_kept )
 * as ONE streaming read of W instead of the L M x M x C products of gpsa_quadform_bwd_alpha (vgpsa.py:192-196
 * and its autograd; the forward alone loses the symmetric half-price form: without a backward use
 * gpsa_quadform_fwd).  Omega [L,M,M] stored as omega_dtype.  workspace >= gpsa_quadform_keep_f32_workspace(M, L).
 * M <= 256: the register-resident full-product kernel; beyond (configs 4 / 5): one tiled product per output into
 * a row-major [L][M][C] buffer, streamed back by a row-blocked kernel - the same entry points, the same sizes
 * queries. */
long long gpsa_quadform_keep_f32_workspace(int M, int L);
long long gpsa_quadform_keep_f32_bytes(int M, long long C, int L);
int gpsa_quadform_fwd_keep_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                               float* v, float* W, void* workspace, long long workspace_bytes, void* stream);
int gpsa_quadform_bwd_alpha_kept_f32(const float* W, const float* g, int M, long long C, int L, const float* dcT,
                                     const float* dmeanT, float* dalpha, void* stream);
/* The data GP's forward, its Gaussian log-likelihood and the backward's abar in ONE pass over the products
 * Omega_l alpha (training with the engine's own ELBO: replaces gpsa_quadform_fwd_keep_f32 + gpsa_data_sample_fwd +
 * gpsa_loglik_fwd/_bwd + gpsa_data_sample_bwd + gpsa_quadform_bwd_alpha_kept_f32; reference vgpsa.py:186-204,
 * 334-351, 532-538 and their autograd adjoints).  Per (l, c), c = s*N + n:
 *   var = (exp(var_u) - q[c]) + alpha_c^T Omega_l alpha_c + 2e-5;  F = meanT[l,c] + sqrt(var) eps[c,l]
 *   z = (Y[n,l] - F) / s,  s = exp(noise_u) + 1e-5;   part[] sums z^2  (LL = -z^2/2 - log s - log(2 pi)/2 per element)
 *   dmeanT[l,c] = dLoss/dF = -(Y - F) / (s^2 S);   g[l,c] = dLoss/dvar = dmeanT * eps / (2 sqrt(var))
 *   abar[:,c]   = 2 sum_l g[l,c] Omega_l alpha_c                (the mean term's share is NOT included)
 * all at upstream gradient dLoss = 1 (linear in it).  part: gpsa_quadform_elbo_parts() doubles (unused tail zeroed).
 * FT (optional, NULL = not wanted): [L][C], the draws themselves, F[s][n][l] at FT[l][s N + n] - for a caller that
 * looks at them afterwards; nothing on the path reads them.
 * M <= 256 (16 row tiles; round 3: 208) only: GPSA_EUNSUPPORTED beyond. */
int gpsa_quadform_elbo_parts(void);
long long gpsa_quadform_elbo_f32_workspace(int M, long long C, int L);
int gpsa_quadform_elbo_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                           const float* meanT, const double* q, const float* var_u, const float* eps, const float* Y,
                           long long N, int S, const float* noise_u, float* g, float* dmeanT, float* abar, double* part,
                           float* FT, void* workspace, long long workspace_bytes, void* stream);
/* The same with the mean formed by the kernel itself: delta [M][L] (delta_F of vgpsa.py:194-196: mean = K_fu K_uu^-1
 * delta = delta^T alpha) is packed into the first padding row of every Omega_l, so that row M of the product
 * Omega_l alpha - MFMAs the padding of the last row tile executes anyway - is delta_l^T alpha_c; no [L,C] mean is read
 * and the caller's [L,M] x [M,C] product goes away.  Only when M is not a multiple of 16 and row M lies in the kernel's
 * last row tile (gpsa_quadform_elbo_takes_delta(M) != 0; M = 200: yes); GPSA_EUNSUPPORTED otherwise. */
int gpsa_quadform_elbo_takes_delta(int M);
int gpsa_quadform_elbo_delta_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                                 const float* delta, const double* q, const float* var_u, const float* eps,
                                 const float* Y, long long N, int S, const float* noise_u, float* g, float* dmeanT,
                                 float* abar, double* part, float* FT, void* workspace, long long workspace_bytes,
                                 void* stream);
/* dOmega[l] = sum_c g[l,c] * alpha[:,c] alpha[:,c]^T  (full symmetric [L,M,M]), stored as out_dtype
 * (out_dtype != dtype only on the fp32 MFMA path, whose partial sums are widened while they are added:
 * GPSA_EUNSUPPORTED otherwise, and the caller converts) */
int gpsa_quadform_bwd_omega(int dtype, int out_dtype, const void* alpha, const void* g, int M, long long C,
                            int L, void* dOmega, void* workspace, long long workspace_bytes, void* stream);
/* The same (fp32 operands, fp32 matrix cores) with the mean term's gradient riding along:
 *   ddelta[m,l] = dbeta * ddelta[m,l] + sum_c alpha[m,c] dmeanT[l,c]      (d delta_F of mean = delta^T alpha, [M][L])
 * comes out of row M of the padded dOmega_l tiles - the first padding row of the kernel's last row tile takes dmeanT[l,c]
 * as its scaled row fragment - so the caller's C-long [M,C] x [C,L] product goes away.  Workspace as
 * gpsa_quadform_workspace(GPSA_F32, M, C, L).  Only where gpsa_quadform_bwd_omega_takes_delta(M, C) != 0 (M not a
 * multiple of 16 and in the kernel's last row tile, M <= 256, C a multiple of 4, 16-byte aligned operands);
 * GPSA_EUNSUPPORTED otherwise. */
int gpsa_quadform_bwd_omega_takes_delta(int M, long long C);
int gpsa_quadform_bwd_omega_delta_f32(int out_dtype, const float* alpha, const float* g, const float* dmeanT, int M,
                                      long long C, int L, void* dOmega, float* ddelta, double dbeta, void* workspace,
                                      long long workspace_bytes, void* stream);

/* alpha = Kinv Kuf on the fp64 matrix cores, with Kinv [M,M] = K_uu^-1 (fp64, from gpsa_chol_inv_f64 +
 * L^-T L^-1) and Kuf [M,C] stored as in_dtype (widened on the fly); alpha [M,C] is stored as alpha_dtype
 * (GPSA_F32 / GPSA_F64); q[c] = sum_m Kuf[m,c] alpha[m,c] (fp64, may be NULL).  These are the
 * K_fu K_uu^-1 factors of the conditional mean and covariance of both GP layers,
 * gpsa/models/vgpsa.py:179-183 and 194-196, and - applied to a gradient panel - the K_uu^-1 solve of
 * their backward.  M <= 256; GPSA_EUNSUPPORTED above (callers then chain gpsa_panel_mm).  workspace >=
 * gpsa_whiten_workspace(M) bytes; the call leaves the matrix-core layout of Kinv in it, and a later call with
 * Kinv == NULL on the same workspace reuses it (the backward's solve against the forward's inverse; the same
 * holds for gpsa_whiten_axpy_f32 and, per problem, gpsa_whiten_batched_f64). */
long long gpsa_whiten_workspace(int M);
int gpsa_whiten_f64(const double* Kinv, int in_dtype, const void* Kuf, int M, long long C,
                    int alpha_dtype, void* alpha, double* q, void* workspace, long long workspace_bytes,
                    void* stream);

/* Y = op(P) X with P [M,M] (stored as p_dtype, op = transpose when transP), X [M,C] and Y in dtype
 * (+ optional colsq[c] = sum_m Y[m,c]^2, may be NULL).
 * The whitening products beta = L^-1 K_uf, alpha = L^-T beta of vgpsa.py:177-180 and their adjoints. */
int gpsa_panel_mm(int dtype, int p_dtype, int transP, const void* P, const void* X, int M, long long C,
                  void* Y, void* colsq, void* workspace, long long workspace_bytes, void* stream);

/* out[m,c] = Y[m,c] + s * d[c] * X[m,c]   (X, Y, out [M,C]; d [C]; out may alias Y).
 * Column-scaled update used by the whitening backward (autograd of vgpsa.py:177-180). */
int gpsa_col_axpy(int dtype, const void* Y, const void* X, const void* d, double s, int M,
                  long long C, void* out, void* stream);

/* ---- reparameterised sampling -----------------------------------------------------------------
 * data GP (vgpsa.py:197-204, 423-426):  var = (exp(var_u) - q[c])_fp64 + v[l,c] + 2e-5 ;
 *   F[c,l] = meanT[l,c] + sqrt(var) * eps[c,l] ;  Sigma[l,c] = var (kept for backward). */
int gpsa_data_sample_fwd(const float* meanT, const float* v, const double* q, const float* var_u,
                         const float* eps, long long C, int L, float* F, float* Sigma, void* stream);
/* given dF [C,L]:  g[l,c] = dF*eps/(2 sqrt(Sigma)),  dmeanT[l,c] = dF[c,l],  qbar[c] = -sum_l g,
 *   dvar_u (device scalar of type dvar_dtype - GPSA_F32 or GPSA_F64 -, overwritten) = exp(var_u) * sum g: the step
 *   engine takes it in fp64 - this share of the data kernel variance's gradient cancels against the covariance
 *   shares to ~1e-3 of its size near a stationary point, and its fp32 rounding alone then shows at 1e-4 of the sum
 *   (round 5).   workspace >= 8*(C/32+2) bytes */
int gpsa_data_sample_bwd(const float* dF, const float* eps, const float* Sigma, const float* var_u,
                         long long C, int L, float* g, float* dmeanT, float* qbar, int dvar_dtype, void* dvar_u,
                         void* workspace, long long workspace_bytes, void* stream);
/* warp GP (vgpsa.py:186-191, 334-351; variance used as the std), fp64 inside, with the
 * linear mean function of the view (vgpsa.py:283-289 mean_slopes / mean_intercepts) evaluated in place:
 *   var = exp(var_u) - q[c] + v[j,c] + 2e-5 ; Gmean[c,j] = (X[c,:] slopes)[j] + intercept[j] + meanT[j,c] ;
 *   Gs[s,c,j] = Gmean[c,j] + var * eps[s,c,j].
 * X [n,D], slopes [D,D], intercept [D], var_u: fp32 as stored.  bad[ceil(n/256)]: per-block flags, 1 if
 * any var <= 0 or NaN (the reference's Normal(...) argument validation raises ValueError there).
 * Gs64 [S,n,D] (may be NULL): the same draws before they are rounded to the fp32 API tensor; the data GP's
 * covariance k(Gtilde, G) (vgpsa.py:409) is evaluated on these, as the reference's fp64 run does. */
int gpsa_warp_sample_fwd(const double* meanT, const double* v, const double* q, const float* var_u,
                         const float* X, const float* slopes, const float* intercept,
                         const float* eps, long long n, int D, int S, float* Gmean, float* Gs,
                         double* Gs64, int* bad, void* stream);
/* given dGmean [n,D] (may be NULL) and dGs [S,n,D] as the sum of an fp32 part dGs and an fp64 part dGs64
 * (either may be NULL: the gradient wrt the fp32 API tensor and wrt its unrounded copy Gs64):
 *   dmeanT[j,c], g[j,c] = sum_s dGs*eps, qbar[c],
 *   dvar_u, dslopes [D,D], dintercept [D] (overwritten).  workspace >= 8*21*ceil(n/256) bytes */
int gpsa_warp_sample_bwd(const float* dGmean, const float* dGs, const double* dGs64, const float* eps,
                         const float* var_u, const float* X, long long n, int D, int S, double* dmeanT, double* g,
                         double* qbar, float* dvar_u, float* dslopes, float* dintercept,
                         void* workspace, long long workspace_bytes, void* stream);

/* linear mean function at the inducing points and the variational residual (vgpsa.py:283-289, 296):
 *   mu_z = scale * (Z slopes + intercept)  [M,D] fp32 ;  resid = delta - mu_z  [M,D] fp64
 * (scale = 100 reproduces the reference's inert x100 on a fixed view) and its adjoint:
 * given dresid: ddelta = dresid, dZ, dslopes, dintercept (all overwritten, fp32). */
int gpsa_mean_resid_fwd(const float* Z, const float* slopes, const float* intercept,
                        const float* delta, int M, int D, double scale, float* mu_z, double* resid,
                        void* stream);
int gpsa_mean_resid_bwd(const double* dresid, const float* Z, const float* slopes, int M, int D,
                        double scale, float* ddelta, float* dZ, float* dslopes, float* dintercept,
                        void* stream);

/* ---- Gaussian likelihood (vgpsa.py:532-538; "variance" used as std) -----------
 * scale = exp(noise_u[0]) + 1e-5 ;  out[0] = sum_{s,n,p} log N(Y[n,p]; F[s,n,p], scale) / S.
 * workspace >= 8*(blocks+2) bytes with blocks = min(4096, ceil(S*N*P/1024)). */
int gpsa_loglik_fwd(const float* F, const float* Y, const float* noise_u, int S, long long N, int P,
                    double* out, void* workspace, long long workspace_bytes, void* stream);
/* dF = gout[0] * dLL/dF ; dnoise_u[0] = gout[0] * dLL/d noise_u  (gout: device scalar, double) */
int gpsa_loglik_bwd(const float* F, const float* Y, const float* noise_u, const double* gout, int S,
                    long long N, int P, float* dF, float* dnoise_u, void* workspace,
                    long long workspace_bytes, void* stream);

/* ---- small helpers used by the KL terms (vgpsa.py:498-530) -------------------------------------
 * out[b] = sum_i A[b,i]*B[b,i] (strideA/strideB in elements; 0 broadcasts) */
int gpsa_bdot(int dtype, const void* A, long long strideA, const void* B, long long strideB,
              long long n, int batch, void* out, void* workspace, long long workspace_bytes,
              void* stream);  /* workspace >= 8*32*batch bytes */
/* A[b] += s * I  (A [batch,M,M]) */
int gpsa_add_diag(int dtype, void* A, int M, int batch, double s, void* stream);

/* ---- KL terms (vgpsa.py:498-530: kl_divergence(MVN(delta_l, Omega_tril_l), MVN(mu_l, Kuu_chol))), fp64 --------
 * kl[l] = 0.5 (logdet K - logdet Omega_l + <K^-1, Omega_l> + d_l^T KD_l - M),  d = delta - mu  [M,L],
 * KD = K^-1 d [M,L] (from gpsa_gemm).  omega_stride / logdet_stride: elements between consecutive l. */
int gpsa_mvn_kl_fwd(const double* Kinv, const double* logdetK, const double* Omega,
                    long long omega_stride, const double* logdetO, long long logdet_stride,
                    const double* Dm, const double* KD, int M, int L, double* kl, void* stream);
/* backward for upstream g[l]:  dOmega[l] = 0.5 g_l (K^-1 - Omega_l^-1) [L,M,M];  dDm = KD diag(g) [M,L];
 * Sp = (sum g) K - sum_l g_l (Omega_l + d_l d_l^T)  [M,M], so that dLoss/dK = 0.5 K^-1 Sp K^-1. */
int gpsa_mvn_kl_bwd(const double* Kuu, const double* Kinv, const double* Omega, long long omega_stride,
                    const double* Oinv, long long oinv_stride, const double* Dm, const double* KD,
                    const double* g, int M, int L, double* dOmega, double* dDm, double* Sp,
                    void* stream);

/* Grouped form: all T KL terms of a step in one launch each way.  The step's matrices live in ONE
 * batch mats [B,M,M] with inverses inv [B,M,M] and logdet [B] (gpsa_chol_inv_f64 + L^-T L^-1): term t
 * pairs the variational covariance mats[om_idx[t]] with the prior mats[pr_idx[t]] (pr_idx[t] < 0: the
 * term is absent - a fixed view - and contributes kl = 0 and zero gradients); D [T,M] holds d_t.
 * fwd: kl [T], KD [T,M] = K_p^-1 d_t (kept for the backward).
 * bwd: the terms are listed per prior: order[grp_off[pg] .. grp_off[pg+1]) are the terms of prior
 *   pr_list[pg], pg < P; group P lists the absent terms.  Writes dOmega [T,M,M], dD [T,M] and
 *   S [P,M,M] with dLoss/dK_p = 0.5 K_p^-1 S_p K_p^-1 (two batched gpsa_gemm calls by the caller). */
int gpsa_mvn_kl_grouped_fwd(const double* mats, const double* inv, const double* logdet,
                            const int* om_idx, const int* pr_idx, const double* D, int M, int T,
                            double* kl, double* KD, void* stream);
int gpsa_mvn_kl_grouped_bwd(const double* mats, const double* inv, const int* om_idx,
                            const int* pr_list, const int* grp_off, const int* order, const double* D,
                            const double* KD, const double* g, int M, int T, int P, double* dOmega,
                            double* dD, double* S, void* stream);

/* ---- ELBO scalar glue (vgpsa.py:540): loss = -(sum_i ll[i]) + kl_scale * sum_t kl[t] ---------------
 * ll [n_ll], kl [n_kl] fp64 device arrays (the per-modality log-likelihoods and the per-term KLs),
 * loss [1] fp32; gpsa_elbo_bwd is the adjoint: dll[i] = -gloss, dkl[t] = kl_scale * gloss. */
int gpsa_elbo_fwd(const double* ll, int n_ll, const double* kl, int n_kl, double kl_scale, float* loss,
                  void* stream);
int gpsa_elbo_bwd(const float* gloss, int n_ll, int n_kl, double kl_scale, double* dll, double* dkl,
                  void* stream);

/* ---- inducing-point initialisation: Lloyd's k-means on the device ----------
 * replaces sklearn.cluster.KMeans at gpsa/models/vgpsa.py:74-76, 90-92.  X [N,D] fp32, centres [K,D].
 * gpsa_kmeans_assign: assign[n] = nearest centre (ties -> lowest index), d2[n] (may be NULL) its
 *   squared distance.   gpsa_kmeans_update: centres[k] = mean of its points (empty cluster: unchanged),
 *   counts[k] (may be NULL).  Deterministic (no atomics). */
long long gpsa_kmeans_workspace(long long N, int D, int K);
int gpsa_kmeans_assign(const float* X, long long N, int D, const float* centres, int K, int* assign,
                       float* d2, void* stream);
int gpsa_kmeans_update(const float* X, const int* assign, long long N, int D, int K, float* centres,
                       int* counts, void* workspace, long long workspace_bytes, void* stream);


/* ---- batched forms: the warp GPs of several views in one launch each way -------------------------------
 * View-blocked layout: the per-view panels of a batch are [M, C] blocks back to back (C = the batch's
 * column stride, >= every view's spot count); problem b has n_live[b] live columns, the rest are zero
 * padding written by the forward (everything computed from a zero column is zero).  fp32 inputs as stored
 * (inducing points, hyper-parameters, coordinates), fp64 arithmetic and results.
 * gpsa_kmat_batched: K[b] = k(Z + b strideZ, X + b strideX) (+ jitter on the diagonal), ls_u / var_u + b
 *   stride_par; n_live (HOST array of batch entries) may be NULL (all C columns live).  batch <= 16.
 * gpsa_kmat_bwd_batched: its adjoint: dZ + b stride_dZ [M,D] and dparams + 2 b (fp64); same != 0: X are the
 *   inducing points themselves (K_uu), the X-side sums are folded into dZ.
 * References as gpsa_kmat / gpsa_kmat_bwd (util.py:8-66 called from vgpsa.py:314-318 for every view). */
int gpsa_kmat_batched(int kind, const float* Z, long long strideZ, int M, const float* X, long long strideX,
                      long long C, int D, const float* ls_u, const float* var_u, int stride_par,
                      const long long* n_live, int batch, double jitter, double* K, long long strideK,
                      void* stream);
long long gpsa_kmat_bwd_batched_workspace(int M, long long C, int D, int batch);
int gpsa_kmat_bwd_batched(int kind, const float* Z, long long strideZ, int M, const float* X, long long strideX,
                          long long C, int D, const float* ls_u, const float* var_u, int stride_par,
                          const long long* n_live, int batch, const double* Kbar, long long strideK, int same,
                          double* dZ, long long stride_dZ, double* dparams, void* workspace,
                          long long workspace_bytes, void* stream);
/* The data GP's covariance backward (autograd of vgpsa.py:409): fp32 Z / hyper-parameters, X = the warp
 * GP's unrounded fp64 draws, an fp32 gradient panel Kbar [M,C]; fp64 arithmetic, partial sums and results
 * (dZ [M,D], dX [C,D] or NULL, dparams[2]).  workspace >= gpsa_kmat_bwd_workspace(GPSA_F64, M, C, D). */
int gpsa_kmat_bwd_x64(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                      const float* var_u, const float* Kbar, double* dZ, double* dX, double* dparams,
                      void* workspace, long long workspace_bytes, void* stream);
/* ... with an fp64 gradient panel (the exact inducing-point gradient of the data GP: gpsa_step_desc.exact_inducing_grad) */
int gpsa_kmat_bwd_x64_f64(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                          const float* var_u, const double* Kbar, double* dZ, double* dX, double* dparams,
                          void* workspace, long long workspace_bytes, void* stream);
/* ... the fp64 panel in two pieces, Kbar[m,c] + s * d[c] * X2[m,c] (X2 [M,C] fp64, d [C] fp32; both NULL: Kbar
 * alone): the exact mode's dK_uf = K^-1 abar + 2 qbar o alpha on the unrounded projection, formed as it is read */
int gpsa_kmat_bwd_x64_f64_axpy(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                               const float* var_u, const double* Kbar, const double* X2, const float* d, double s,
                               double* dZ, double* dX, double* dparams, void* workspace, long long workspace_bytes,
                               void* stream);
/* C += -(G + d o A) A^T: the exact mode's dK_uu = -(K^-1 abar + qbar o alpha) alpha^T (autograd of vgpsa.py:177-180
 * through K_uu) as ONE C-long fp64 product whose left operand is formed from its two panels as it is staged.
 * G, A [M, C] fp64, d [C] fp32, dK [M, M] fp64 (added to); workspace >= gpsa_exact_dkuu_workspace(M, C) bytes. */
long long gpsa_exact_dkuu_workspace(int M, long long C);
int gpsa_exact_dkuu_f64(const double* G, const double* A, const float* d, int M, long long C, double* dK,
                        void* workspace, long long workspace_bytes, void* stream);
/* Long-K fp64 products with a small square result, nprob of them in one launch (csrc/longk64.hip):
 *     out[p] [M, M] = beta[p] out[p] + alpha[p] * sum_k (G[p][:, k] + d[p][k] B[p][:, k]) B[p][:, k]^T
 * G[p], B[p]: [M, K] row-major fp64 panels with leading dimension ld (G == NULL: the left operand is d o B alone);
 * d[p]: [K] of type d_dtype (GPSA_F32 / GPSA_F64; d == NULL: the left operand is G alone); sym != 0: the caller
 * guarantees a symmetric result (the lower triangle is computed and mirrored).  The step's products of this shape:
 * the warp GPs' dOmega_j = sum_c g_j alpha alpha^T and dK_uu = -(gamma + qbar o alpha) alpha^T, and the exact
 * inducing-point gradient's dK_uu of the data GP (autograd of vgpsa.py:177-196 through K_uu and Omega).  alpha, beta,
 * and the pointer arrays are HOST arrays of nprob entries.  gpsa_longk_f64_workspace == 0 / GPSA_EUNSUPPORTED: the
 * shape is not covered (M > 256, odd or short K, unaligned panels) - use gpsa_gemm. */
long long gpsa_longk_f64_workspace(int M, long long K, int nprob);
int gpsa_longk_f64(int nprob, const double* const* G, const double* const* B, const void* const* d, int d_dtype,
                   int M, long long K, long long ld, int sym, const double* alpha, const double* beta,
                   double* const* out, void* workspace, long long workspace_bytes, void* stream);
/* ... with the gradient panel in two pieces, Kbar[m,c] + s * d[c] * X2[m,c] (X2 [M,C], d [C] fp32; both NULL: Kbar
 * alone): the data GP's dK_uf = K^-1 abar + 2 qbar o alpha (autograd of vgpsa.py:177-196) formed as it is read */
int gpsa_kmat_bwd_x64_axpy(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                           const float* var_u, const float* Kbar, const float* X2, const float* d, double s,
                           double* dZ, double* dX, double* dparams, void* workspace, long long workspace_bytes,
                           void* stream);
/* gpsa_whiten_f64 on an fp64 panel with the result stored twice from the same accumulators: unrounded (alpha64) and
 * rounded to fp32 (alpha32: what the matrix-core contractions read) - the exact inducing-point gradient of the data
 * GP keeps both (gpsa_step_desc.exact_inducing_grad; autograd of vgpsa.py:177-180, 409) */
int gpsa_whiten_f64_dual(const double* Kinv, const double* Kuf, int M, long long C, double* alpha64, float* alpha32,
                         double* q, void* workspace, long long workspace_bytes, void* stream);

/* gpsa_whiten_f64_dual with K_uf[m, c] = k(Z_m, x_c) formed inside the projection kernel (Z [M,D] fp32, X64 [C,D]
 * fp64, log-parameters fp32: what gpsa_kmat(GPSA_F64, GPSA_F32_X64, ...) evaluates) - vgpsa.py:171-189 in one pass,
 * no K_uf in memory.  GPSA_EUNSUPPORTED for shapes the persistent kernel does not take (the caller then runs
 * gpsa_kmat + gpsa_whiten_f64_dual; GPSA_PROJ64_GEN=0 forces that). */
int gpsa_whiten_gen_f64_dual(const double* Kinv, int kind, const float* Z, const double* X64, int D, const float* ls_u,
                             const float* var_u, int M, long long C, double* alpha64, float* alpha32, double* q,
                             void* workspace, long long workspace_bytes, void* stream);
/* gpsa_whiten_f64 on an fp32 panel with gpsa_col_axpy fused into its store:
 *   out[m,c] = (Kinv X)[m,c] + s * d[c] * X2[m,c]      (X, X2, out [M,C] fp32; d [C] fp32; fp64 arithmetic)
 * the data GP's dK_uf = K^-1 abar + 2 qbar o alpha (autograd of vgpsa.py:177-196) in one pass. */
int gpsa_whiten_axpy_f32(const double* Kinv, const float* X, int M, long long C, const float* X2, const float* d,
                         double s, float* out, void* workspace, long long workspace_bytes, void* stream);
/* gpsa_whiten_f64 for a batch of fp64 panels with one inverse each: problem b reads Kinv + b strideKinv and
 * Kuf + b strideX, writes alpha + b strideX and q + b C (q may be NULL).
 * workspace >= batch * gpsa_whiten_workspace(M). */
int gpsa_whiten_batched_f64(const double* Kinv, long long strideKinv, const double* Kuf, int M, long long C,
                            long long strideX, double* alpha, double* q, int batch, void* workspace,
                            long long workspace_bytes, void* stream);
/* gpsa_quadform_fwd_keep / _bwd_alpha_kept / gpsa_col_axpy (fp64) for ``batch`` layers with contiguous
 * operands: alpha [batch][M,C], Omega [batch][L,M,M], W [batch][L,M,C], v / meanT / g / dmeanT [batch][L,C],
 * dcT [batch][M,L], d [batch][C]. */
int gpsa_quadform_fwd_keep_batched_f64(const double* alpha, const double* Omega, int M, long long C, int L,
                                       double* v, double* W, const double* dcT, double* meanT, int batch,
                                       void* stream);
int gpsa_quadform_bwd_alpha_kept_batched_f64(const double* W, const double* g, int M, long long C, int L,
                                             const double* dcT, const double* dmeanT, double* dalpha,
                                             int batch, void* stream);
int gpsa_col_axpy_batched_f64(const double* Y, const double* X, const double* d, double s, int M, long long C,
                              double* out, int batch, void* stream);
/* dOmega[b][l] = sum_c g[b][l,c] alpha[b][:,c] alpha[b][:,c]^T (fp64, ``batch`` layers with contiguous
 * operands; gpsa_quadform_bwd_omega for the views' warp GPs in one launch sequence). */
long long gpsa_gram_batched_workspace(int M, long long C, int L, int batch);
int gpsa_gram_batched_f64(const double* alpha, const double* g, int M, long long C, int L, double* dOmega,
                          int batch, void* workspace, long long workspace_bytes, void* stream);
/* gpsa_mvn_kl_grouped_bwd with accumulate != 0: dOmega += ... (it already holds the layers' share). */
int gpsa_mvn_kl_grouped_bwd_acc(const double* mats, const double* inv, const int* om_idx,
                                const int* pr_list, const int* grp_off, const int* order, const double* D,
                                const double* KD, const double* g, int M, int T, int P, double* dOmega,
                                double* dD, double* S, int accumulate, void* stream);

/* ==== the step engine: forward(S) and its backward as ONE host call each =================================
 * Replaces the body of VariationalGPSA.forward (gpsa/models/vgpsa.py:212-489) plus the KL terms of loss_fn
 * (vgpsa.py:498-530), and what autograd derives from them, for the built-in covariance functions: the host
 * enqueues the whole launch sequence of a step from C++ (no per-launch Python / ctypes round trip), the
 * warp GPs of all free views run batched in the same launches, every parameter gradient is accumulated in
 * fp64 along all of its paths and rounded to the fp32 parameter once.
 *
 * A plan (gpsa_step_create) fixes the problem's shape; it owns only host tables and a few hundred bytes of
 * device index tables (allocated at creation - the ONLY allocation of the library; the hot entry points
 * below allocate nothing, never synchronise, and are safe to capture into a hipGraph).  The caller passes
 * two arenas per call: ``saved`` (gpsa_step_saved_bytes: what the backward needs from the forward; one per
 * live forward) and ``scratch`` (gpsa_step_scratch_bytes: transient, reusable by the next call on the same
 * stream).
 *
 * Views are consecutive row blocks of each modality (what GPSA.create_view_idx_dict produces,
 * gpsa/models/gpsa.py:155-183); spots of all modalities of a view share that view's warp GP
 * (vgpsa.py:284-294).  All parameters are the fp32 tensors of the reference's state_dict. */
#define GPSA_MAX_MODS 4

typedef struct gpsa_step_desc {
  int n_views, n_dims, n_mods, n_samples;   /* V, D (<= 4), modalities (<= GPSA_MAX_MODS), S */
  int m_x, m_g;                             /* inducing points per view / of the data GP */
  int kind_warp, kind_data;                 /* GPSA_K_* */
  int n_latent[GPSA_MAX_MODS];              /* L_m: latent outputs of the data GP (= P_m without LMC) */
  int n_out[GPSA_MAX_MODS];                 /* P_m */
  int has_lmc[GPSA_MAX_MODS];               /* F_obs = F_latent W (vgpsa.py:428-432) */
  long long n_rows[GPSA_MAX_MODS];          /* N_m */
  int s_test;                               /* leading dim of G_test, 0: no test pass (vgpsa.py:437-477) */
  long long n_test[GPSA_MAX_MODS];          /* rows of G_test[m] */
  int want_kl;                              /* factorise the variational covariances and evaluate the KL terms */
  const int* view_fixed;                    /* HOST [V]: 1 = fixed view (vgpsa.py:262-273) */
  const long long* view_rows;               /* HOST [n_mods * V]: rows of view v in modality m, [m * V + v] */
  long long keep_budget_bytes;              /* HBM a training forward may spend on the data GPs' kept products
                                               Omega_l alpha (L_m m_g C floats per pass): > 0 that many bytes; < 0
                                               never keep; 0: GPSA_KEEP_GB (default 48) GiB, and no more than 60 % of
                                               the device memory free at gpsa_step_create */
  int exact_inducing_grad;                  /* 1: the data GP's gradient wrt its inducing points Gtilde from the UNROUNDED
                                               projection: alpha = K^-1 K_uf kept in fp64 for the backward, gamma =
                                               K^-1 abar stored in fp64, dK_uu = -(gamma + qbar alpha) alpha^T as one
                                               C-long fp64 product.  The K_uu and K_uf shares of that gradient cancel to
                                               1e-4 .. 1e-5 of their size, so fp32 roundings of alpha / gamma are 1e-3 of
                                               the result at M >= 200.  +8 M C bytes, one M x M x C fp64 product */
  int kl_own_lo, kl_own_hi;                 /* data-parallel ranks, OWNER COMPUTES: this plan evaluates only the KL terms
                                               kl_own_lo <= t < kl_own_hi of the V*D + sum L_m terms (order: Omega_G rows
                                               r = j*V+v, then every modality's outputs; vgpsa.py:498-530) - the other
                                               terms come out as 0 with zero gradients and their variational covariances
                                               are neither factorised nor inverted.  kl_own_hi <= 0: every term (one
                                               process, or the 1/world weighting).  Summed over ranks whose ranges
                                               partition the terms, loss and gradients are the full ELBO's. */
} gpsa_step_desc;

typedef struct gpsa_step_params {           /* device pointers, fp32, the reference's parameter layout */
  const float* Xtilde;                      /* [V, m_x, D] */
  const float* delta_G;                     /* [V, m_x, D] */
  const float* Omega_sqt_G;                 /* [V*D, m_x, m_x], row j*V+v (vgpsa.py:131-143) */
  const float* warp_ls;                     /* [V] log lengthscales */
  const float* warp_var;                    /* [V] log variances */
  const float* slopes;                      /* [V, D, D] */
  const float* intercepts;                  /* [V, D] */
  const float* Gtilde;                      /* [m_g, D] */
  const float* data_ls;                     /* [1] */
  const float* data_var;                    /* [1] */
  const float* Omega_sqt_F[GPSA_MAX_MODS];  /* [L_m, m_g, m_g] */
  const float* delta_F[GPSA_MAX_MODS];      /* [m_g, L_m] */
  const float* W[GPSA_MAX_MODS];            /* [L_m, P_m] or NULL */
} gpsa_step_params;

typedef struct gpsa_step_param_grads {      /* fp32 outputs, overwritten; NULL = not wanted */
  float *Xtilde, *delta_G, *Omega_sqt_G, *warp_ls, *warp_var, *Gtilde, *data_ls, *data_var;
  float* Omega_sqt_F[GPSA_MAX_MODS];
  float* delta_F[GPSA_MAX_MODS];
  float* W[GPSA_MAX_MODS];
} gpsa_step_param_grads;

typedef struct gpsa_step_io {
  const float* X[GPSA_MAX_MODS];            /* in  [N_m, D] spatial coordinates */
  const float* eps_G;                       /* in  standard-normal draws of the free, non-empty views back to
                                                   back, each [S, n_v, D] (order of vgpsa.py:346-348) */
  const float* eps_F[GPSA_MAX_MODS];        /* in  [S, N_m, L_m] (vgpsa.py:423) */
  const float* G_test[GPSA_MAX_MODS];       /* in  [s_test, n_test_m, D] or NULL */
  const float* eps_F_test[GPSA_MAX_MODS];   /* in  [s_test, n_test_m, L_m] */
  float* G_means[GPSA_MAX_MODS];            /* out [N_m, D] */
  float* G_samples[GPSA_MAX_MODS];          /* out [S, N_m, D] */
  float* F_latent[GPSA_MAX_MODS];           /* out [S, N_m, L_m] */
  float* F_obs[GPSA_MAX_MODS];              /* out [S, N_m, P_m] with LMC, else NULL (F_obs is F_latent) */
  float* F_latent_test[GPSA_MAX_MODS];      /* out [s_test, n_test_m, L_m] */
  float* F_obs_test[GPSA_MAX_MODS];
  float* mu_z;                              /* out [V, m_x, D]: prior means at the inducing points (mu_z_G) */
  double* kl;                               /* out [gpsa_step_n_kl] per-term KL (order: Omega_G rows r = j*V+v,
                                                   then every modality's outputs), NULL with want_kl = 0 */
  int* flag;                                /* out [1]: nonzero = a covariance was not positive definite or a
                                                   warp variance not positive (the reference raises there) */
  int keep_products;                        /* in  nonzero (training): the data GPs' forward keeps its products
                                                   Omega_l alpha in the saved arena (gpsa_quadform_fwd_keep_f32)
                                                   and gpsa_step_backward streams them back; needs the arena of
                                                   gpsa_step_saved_bytes.  0 (no backward to follow): the cheaper
                                                   forward, gpsa_step_saved_bytes_nokeep suffices, and a backward
                                                   recomputes the products.  Pass the same value to both calls. */
  int reuse_mm;                             /* in  nonzero: the saved arena still holds the M x M stage (prior and
                                                   variational covariances, their factorisations and inverses, KL terms)
                                                   of an earlier gpsa_step_forward on the SAME parameters - skip it.
                                                   For the slices of one microbatched step (train.Microbatches) */
  /* Fused ELBO (training through the engine's own loss, Gaussian likelihood on F_latent, no LMC): with fuse_elbo
   * nonzero and Y[m] / noise_u[m] given, modality m's data GP runs gpsa_quadform_elbo_f32 - F_latent[m] is not
   * written (the draws leave only through the optional F_fused_T[m]), ll_part[m] receives the partial sums of z^2 for
   * gpsa_elbo_loss_fused_fwd / _bwd, and gpsa_step_backward takes the loss's upstream gradient from
   * gpsa_step_out_grads.gloss instead of dF_latent[m].  Modalities the fused kernel does not cover (LMC, more than
   * 256 inducing points: gpsa_step_fused(plan, m) == 0) run unfused and must be given F_latent[m] as usual. */
  int fuse_elbo;
  const float* Y[GPSA_MAX_MODS];            /* in  [N_m, L_m] observations */
  const float* noise_u[GPSA_MAX_MODS];      /* in  [1] the likelihood's log "variance" of modality m */
  double* ll_part[GPSA_MAX_MODS];           /* out [gpsa_quadform_elbo_parts()] */
  float* F_fused_T[GPSA_MAX_MODS];          /* out, optional: [L_m][S N_m] the draws of a fused modality, F[s][n][l] at
                                                   [l][s N_m + n] (gpsa_quadform_elbo_f32's FT) */
  /* One optimiser step as several forward / backward passes over row slices (train.Microbatches) that CLOSE once:
   * bwd_acc = gpsa_step_bwd_acc_bytes(plan) bytes the caller keeps across the slices' gpsa_step_backward calls;
   * bwd_acc_mode 1: first slice (its N-scaled gradient pieces are stored there, nothing else is done: ``grads`` is not
   * written), 2: a middle slice (added), 3: the last slice (the accumulator is added to its own pieces, then the M x M
   * closing - KL backward, prior covariances' backward, dOmega -> dA, finalisation - runs once and ``grads`` is written),
   * 0: an ordinary backward.  The KL terms' gradient og->dkl belongs to the LAST slice's call. */
  double* bwd_acc;
  int bwd_acc_mode;
  /* Data-parallel overlap (round 5; parallel.GradAllReducer(overlap=True)): with f_event (a hipEvent_t) given,
   * gpsa_step_backward finishes the DATA GP's span of the gradients first - Omega_sqt_F, delta_F, W: 97 % of the bytes
   * at the headline configuration - and records the event on the caller's stream behind it, so that the caller can
   * start reducing that span on another stream while the warp GPs' backward, the priors' covariance backward and the
   * rest run.  The KL backward then runs in front of the warp GPs' backward, whose products are added to its shares
   * instead of overwriting them: the fp64 sums meet in another order (results equal to rounding, not bitwise).  The
   * event is recorded at the very end when the early order is not available (a microbatched step, the side stream, a
   * shape the long-K kernel does not cover).  NULL: the ordinary order. */
  void* f_event;
} gpsa_step_io;

typedef struct gpsa_step_out_grads {        /* gradients of the caller's scalar wrt the forward's outputs */
  const float* dG_means[GPSA_MAX_MODS];     /* each may be NULL (= zero) */
  const float* dG_samples[GPSA_MAX_MODS];
  const float* dF_latent[GPSA_MAX_MODS];
  const float* dF_obs[GPSA_MAX_MODS];
  const float* dF_latent_test[GPSA_MAX_MODS];
  const float* dF_obs_test[GPSA_MAX_MODS];
  const double* dkl;                        /* [n_kl] or NULL */
  const float* gloss;                       /* [1] device scalar: upstream gradient of the loss (io.fuse_elbo) */
} gpsa_step_out_grads;

void* gpsa_step_create(const gpsa_step_desc* desc);   /* NULL: invalid / unsupported description */
/* the same plan described on the host only (no device): out[7] = saved bytes, scratch bytes, KL terms, floats of
 * eps_G, batched runs of free views, column stride of the view blocks, saved bytes without the kept products */
int gpsa_step_describe(const gpsa_step_desc* desc, long long* out);
void gpsa_step_destroy(void* plan);
long long gpsa_step_saved_bytes(const void* plan);
long long gpsa_step_saved_bytes_nokeep(const void* plan);  /* arena of a forward with io.keep_products == 0 */
int gpsa_step_fused(const void* plan, int m);         /* 1: modality m's training pass can run the fused ELBO kernel */
long long gpsa_step_scratch_bytes(const void* plan);
long long gpsa_step_bwd_acc_bytes(const void* plan);  /* gpsa_step_io.bwd_acc */
int gpsa_step_n_kl(const void* plan);                 /* V*D + sum_m L_m */
int gpsa_step_n_factorised(const void* plan);         /* matrices a forward with want_kl factorises: the priors + the
                                                         variational covariances of the plan's own KL terms
                                                         (gpsa_step_desc.kl_own_lo / _hi; all of them by default) */
long long gpsa_step_eps_g_numel(const void* plan);    /* floats in gpsa_step_io.eps_G */
/* EXPERIMENTAL, off by default: gpsa_step_forward / _backward can replay a cached hipGraph of their launch sequence
 * when they meet an argument set (pointer structs, arenas, stages, stream) for the second time, instead of enqueueing
 * 20 - 45 launches one by one (csrc/step.hip: "hipGraph cache" has the measurements and an open issue).  For callers
 * that own fixed buffers; GPSA_STEP_GRAPH=1 turns it on for the process.  This is the cache's switch and its counters:
 * enable != 0 / 0 (-1: leave as it is); out[0..3] = replays, eager calls, captures, graphs held (out may be NULL). */
int gpsa_step_graph(void* plan, int enable, long long* out);
/* backwards of this plan that took the early order of gpsa_step_io.f_event (the data GP's gradients finished first) */
long long gpsa_step_early_backwards(const void* plan);
/* Where the step's factorisation batch sits in the ``saved`` arena, for the reference's forward -> loss_fn hand-off
 * attributes Kuu_chol_list / curr_Omega_tril_list / Kuu_chol_F / curr_Omega_tril_F (vgpsa.py:237, 257, 321, 394, 412):
 * the engine keeps the matrices (K_uu + 1e-5 I, Omega = A A^T + 1e-5 I; fp64, [B, M, M] per group) and their inverses,
 * not the Cholesky factors, so the attributes are formed from these on access.  out[0] = number of groups (1: m_x ==
 * m_g, one batch; 2: the warp GPs' group, then the data GP's); per group g at out[1 + 4 g]: M, priors, variational
 * covariances, byte offset of the batch.  Order within a group: the priors (free views in order; the data GP's last /
 * alone), then Omega_G rows 0 .. V*D-1 (group 0), then every modality's Omega_F rows. */
int gpsa_step_batch_layout(const void* plan, long long* out);
/* stages: bit 0 = the M x M factorisations, the KL terms and the warp GPs (everything ``flag`` depends on),
 * bit 1 = the data GPs.  3 = the whole forward; a caller that wants to look at ``flag`` while the data GPs
 * run enqueues the two stages with two calls and its flag copy in between.
 * bits 8 .. 8 + GPSA_MAX_MODS - 1 (with bit 1): run the data GP of these modalities' training rows only (0: every
 * pass), on the ``saved`` arena stage 1 filled.  This is how the reference's two calls map onto the fused ELBO:
 * forward() enqueues stage 1 (+ the modalities that cannot fuse), loss_fn(), which is where the observations arrive
 * (vgpsa.py:491, 532-538), sets io.Y[m] and enqueues modality m's pass.  A caller that wants modality m's draws after
 * all passes Y[m] = NULL and F_latent[m] instead (the unfused kernels; the same io then goes to gpsa_step_backward).
 * bit 2: such a pass is an extra (re-)run, not one of the step's launches gpsa_step_timing records. */
int gpsa_step_forward(void* plan, const gpsa_step_params* params, const gpsa_step_io* io, void* saved,
                      void* scratch, int stages, void* stream);
/* diagnostic (bench.py's roofline figures): HIP events around the three contraction launches (the variance
 * form, its alpha-gradient, its Omega-gradient) of the first data-GP pass, on the stream they are launched on,
 * for a ring of ``slots`` steps; gpsa_step_timing_read returns the recorded steps' durations [n][3] in ms
 * (after the caller synchronised).  slots = 0 switches the events off. */
/* id of the stream capture ``stream`` belongs to (0: not capturing).  For hosts that cache scratch per stream: a block
 * allocated inside a capture lives in that graph's private pool and must not be reused by another capture. */
unsigned long long gpsa_stream_capture_id(void* stream);
int gpsa_step_timing(void* plan, int slots);
int gpsa_step_timing_read(void* plan, float* ms, int max_steps);
/* backward of the forward that filled ``saved`` (same params / io pointers and contents) */
int gpsa_step_backward(void* plan, const gpsa_step_params* params, const gpsa_step_io* io,
                       const gpsa_step_out_grads* og, void* saved, void* scratch,
                       const gpsa_step_param_grads* grads, void* stream);

/* ---- Gaussian likelihood + ELBO in one call each way (vgpsa.py:532-540) ------------------------------------
 * loss[0] = -(sum_i LL_i) + kl_scale * sum_t kl[t],  LL_i = sum log N(Y_i; F_i, exp(noise_u[i]) + 1e-5) / S_i
 * (n_ll likelihood terms: one per modality; F_i [S, N_i, P_i], Y_i [N_i, P_i]; kl [n_kl] fp64 or NULL).
 * backward: dF_i (overwritten), dnoise[i] (overwritten), dkl[t] = kl_scale * gloss; dnoise_all [n_noise] or NULL:
 * the whole noise-gradient vector the dnoise[i] point into, zero-filled first (the reference's noise_variance
 * holds entries no likelihood term reads, vgpsa.py:217 / :534).
 * workspace >= 8 * 4100 * n_ll bytes. */
int gpsa_elbo_loss_fwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                       const int* S, const long long* N, const int* P, const double* kl, int n_kl,
                       double kl_scale, float* loss, double* ll_out, void* workspace,
                       long long workspace_bytes, void* stream);
int gpsa_elbo_loss_bwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                       const int* S, const long long* N, const int* P, const float* gloss, int n_kl,
                       double kl_scale, float* const* dF, float* const* dnoise, float* dnoise_all, int n_noise,
                       double* dkl, void* workspace, long long workspace_bytes, void* stream);

/* gpsa_elbo_loss_fwd / _bwd with some likelihood terms fused into the step (gpsa_step_io.fuse_elbo): zpart[i] non-null =
 * term i's partial sums of z^2 (nparts doubles: gpsa_step_io.ll_part[m]); F[i] / dF[i] are then ignored */
int gpsa_elbo_loss_fused_fwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                             const int* S, const long long* N, const int* P, const double* const* zpart, int nparts,
                             const double* kl, int n_kl, double kl_scale, float* loss, double* ll_out, void* workspace,
                             long long workspace_bytes, void* stream);
int gpsa_elbo_loss_fused_bwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                             const int* S, const long long* N, const int* P, const double* const* zpart, int nparts,
                             const float* gloss, int n_kl, double kl_scale, float* const* dF, float* const* dnoise,
                             float* dnoise_all, int n_noise, double* dkl, void* workspace, long long workspace_bytes,
                             void* stream);
/* ---- LMC likelihood without F_obs (round 4; replaces, for an LMC modality in training, the product
 * F_obs = F_latent W of vgpsa.py:428-432, the Gaussian likelihood over [S, N, P] of vgpsa.py:532-538 and autograd's
 * dF_latent = dF_obs W^T, dW = F_latent^T dF_obs): one pass that forms, at upstream gradient 1 of loss = -LL,
 *   zpart[]  block partials of sum ((Y[n,p] - F_obs[s,n,p]) / s)^2  (nparts doubles, tail zeroed: the ``zpart`` of
 *            gpsa_elbo_loss_fused_fwd / _bwd, which finish LL and the noise gradient from it),
 *   dF [S N, L] = dLoss/dF_latent,   dW [L, P] = dLoss/dW,        with s = exp(noise_u) + 1e-5, dF_obs = -(Y - F_obs)/(s^2 S).
 * F [S N, L] (F_latent, sample-major as the API tensor), W [L, P], Y [N, P].  L <= 64 (GPSA_EUNSUPPORTED beyond); since
 * round 5 the three products run on the matrix cores (csrc/lmc.hip; GPSA_LMC_MFMA=0: the vector-pipe kernel, L <= 32). */
long long gpsa_lmc_loglik_workspace(long long C, int L, int P, int nparts);
int gpsa_lmc_loglik_fused_f32(const float* F, const float* W, const float* Y, const float* noise_u, int S, long long N,
                              int L, int P, double* zpart, int nparts, float* dF, float* dW, void* workspace,
                              long long workspace_bytes, void* stream);
/* the fused forward's outputs (formed at upstream gradient 1) as the backward wants them: g, dmeanT, abar scaled by the
 * loss's upstream gradient (untouched when it is 1), g_ext row L = qbar = -sum_l g, dvar_u = exp(var_u) sum g */
int gpsa_elbo_fused_post(float* g_ext, float* dmeanT, float* abar, int M, long long C, int L, const float* gloss,
                         const float* var_u, int dvar_dtype, void* dvar_u, void* workspace, long long workspace_bytes,
                         void* stream);

/* ---- fused Adam over a list of tensors (torch.optim.Adam, no weight decay / amsgrad; the optimiser step of
 * the reference loop, examples/grid_example.py:59-78), ONE launch: for every element
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * with t = ++step[0] kept on the device (capturable).  ptrs: HOST arrays of n device pointers. */
int gpsa_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const long long* numel, double lr, double beta1, double beta2,
                   double eps, float* step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPSA_HIP_H */
