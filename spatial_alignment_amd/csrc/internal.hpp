// Launchers shared between the translation units of libgpsa_hip (defined where their kernels live).
#pragma once
#include "common.hpp"

namespace gpsa {

// gemm.hip: C[b] = alpha op(A[b]) op(B[b]) + beta C[b], row-major, strided batch, deterministic split-K
template <typename T>
int gemm_launch(int transA, int transB, int m, int n, long long k, double alpha, const T* A, long long lda,
                long long sA, const T* B, long long ldb, long long sB, double beta, T* C, long long ldc,
                long long sC, int batch, int splitk, void* ws, long long ws_bytes, hipStream_t st);
// triangle modes of the MFMA product (gemm.hip documents them at the kernel)
enum { GEMM_TRI_NONE = 0, GEMM_TRI_UPPER_A = 1, GEMM_TRI_LOWER_C = 2, GEMM_TRI_LTL = 3 };
template <typename T>
int gemm_launch_tri(int transA, int transB, int m, int n, long long k, double alpha, const T* A, long long lda,
                    long long sA, const T* B, long long ldb, long long sB, double beta, T* C, long long ldc,
                    long long sC, int batch, int splitk, void* ws, long long ws_bytes, hipStream_t st, int tri);
// the same accumulated in fp64 from operands / into results of either precision (instantiated for
// (double,float,float), (float,float,double), (double,double,float))
template <typename TIA, typename TIB, typename TO>
int gemm64_launch(int transA, int transB, int m, int n, long long k, double alpha, const TIA* A, long long lda,
                  long long sA, const TIB* B, long long ldb, long long sB, double beta, TO* C, long long ldc,
                  long long sC, int batch, int splitk, void* ws, long long ws_bytes, hipStream_t st);

// quadform.hip: Omega = A A^T + jitter I with LDS-DMA staging (GPSA_EUNSUPPORTED: shape / alignment not covered)
int omega_fwd_dma_launch(const float* A0, int n0, double* O0, const float* A1, int n1, double* O1, int M, double jitter,
                         hipStream_t st);

// quadform.hip: dA = 2 G A for symmetric G with the same staging
int omega_bwd_dma_launch(const double* G0, const float* A0, float* D0, int n0, const double* G1, const float* A1,
                         float* D1, int n1, int M, hipStream_t st);

// kl.hip: gpsa_mvn_kl_grouped_fwd that also writes the terms to kl_copy (may be NULL)
int mvn_kl_grouped_fwd_slices(int M);  // row slices per term (sizes the partial sums: [T][slices] doubles)
int mvn_kl_grouped_fwd_copy(const double* mats, const double* inv, const double* logdet, const int* om_idx,
                            const int* pr_idx, const double* D, int M, int T, double* kl, double* KD,
                            double* kl_copy, double* part, int* cnt, hipStream_t st);

// lmc.hip: the LMC likelihood, its gradient and the two LMC gradient products on the matrix cores, G workgroups
// (zpart [nparts >= G], dWpart [G][L][P]); GPSA_EUNSUPPORTED beyond 64 latent outputs
int lmc_mfma_launch(const float* F, const float* W, const float* Y, const float* noise_u, int S, long long N, int L,
                    int P, double* zpart, int nparts, float* dF, float* dWpart, int G, hipStream_t st);

// proj64.hip: alpha = Kinv X (fp64 matrix cores) as a persistent output-stationary kernel over pack_whiten_kernel's
// packed inverse(s); proj64_ok says which shapes it takes (GPSA_PROJ64=0: none); q is closed by atomic adds onto
// zero: q_zeroed says an earlier launch on the stream has cleared it (else a memset node goes in front)
bool proj64_ok(int MB, long long C, int batch);
template <typename TI>
int proj64_launch(int MB, const double* Apk, const TI* X, int M, long long C, double* alpha, float* out32, double* q,
                  int batch, long long sX, hipStream_t st, bool q_zeroed);
// zero fill as a kernel of this library, not a runtime memset (elementwise.hip: a memset NODE of a replayed graph was
// seen to land behind the kernel that follows it); returns 0 or a hipError_t
int zero_fill_async(void* ptr, size_t bytes, hipStream_t st);
int copy_async(void* dst, const void* src, size_t bytes, hipStream_t st);  // device to device, as a kernel
// gpsa_elbo_fused_post as one launch (elementwise.hip): tick = a device word that is zero at entry
int elbo_fused_post_ticket(float* g_ext, float* dmeanT, float* abar, int M, long long C, int L, const float* gloss,
                           const float* var_u, double* dvar_u, double* part, int* tick, hipStream_t st);
// the same with the right-hand side K_uf[m, c] = k(Z_m, x_c) formed inside the kernel (round 6)
int proj64_gen_launch(int MB, const double* Apk, int kind, const float* Z, const double* X64, int D, const float* ls_u,
                      const float* var_u, int M, long long C, double* alpha, float* out32, double* q, hipStream_t st,
                      bool q_zeroed);

}  // namespace gpsa
