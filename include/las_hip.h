/* liblas_hip.so  --  C ABI of the MI355X-native LAS training hot path.
 *
 * The reference (Chung-I/End-to-end-ASR-Pytorch) has no FFI of its own: it delegates every
 * operator below to PyTorch (cuDNN/cuBLAS/ATen).  Each entry point names the reference call
 * site it replaces (paths under the reference repo).  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked host;
 *   - the caller owns all buffers, including workspaces (query las_*_workspace_bytes first);
 *     the library never allocates or frees device memory and never synchronises the stream;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, <0 = LAS_E_* (bad argument / unsupported shape), >0 = hipError_t;
 *   - fp32 storage everywhere; `prec` selects the MFMA operand format of GEMM-shaped work:
 *     LAS_PREC_BF16 (bf16 operands, fp32 accumulate) or LAS_PREC_F32 (exact f32 MFMA).
 */
#ifndef LAS_HIP_H
#define LAS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LAS_ABI_VERSION 1
enum { LAS_OK = 0, LAS_E_BADARG = -1, LAS_E_UNSUPPORTED = -2, LAS_E_WORKSPACE = -3, LAS_E_TIMEOUT = -4 };
enum { LAS_PREC_BF16 = 0, LAS_PREC_F32 = 1 };
enum { LAS_ACT_NONE = 0, LAS_ACT_TANH = 1 };

int las_abi_version(void);
const char* las_error_string(int code);

/* ---- CTC loss with fused log-softmax ------------------------------------------------------
 * Replaces F.log_softmax(ctc_pred.transpose(0,1),-1) + torch.nn.CTCLoss(blank=0) at
 * src/solver.py:93,160 (and :253).  logits [B,T,V] raw, batch-major as src/asr.py:69 emits them.
 * label [B,L] zero padded (includes <eos>=1); enc_len/tgt_len [B] int32.
 * fwd: nll [B] (+inf if infeasible), log_alpha [B,T,2L+1] (-inf outside the lattice).
 * bwd: grad_logits [B,T,V] = gscale[b] * d nll_b / d logits  (NaN rows where nll=+inf, as ATen).
 * The same workspace must be passed to bwd after fwd (it carries the row log-sum-exps). */
size_t las_ctc_workspace_bytes(int B, int T, int V, int L);
int las_ctc_loss_fwd(const float* logits, const int32_t* label, const int32_t* enc_len, const int32_t* tgt_len,
                     int B, int T, int V, int L, int blank, float* nll, float* log_alpha,
                     void* workspace, size_t ws_bytes, void* stream);
int las_ctc_loss_bwd(const float* logits, const int32_t* label, const int32_t* enc_len, const int32_t* tgt_len,
                     int B, int T, int V, int L, int blank, const float* nll, const float* log_alpha,
                     const float* gscale, float* grad_logits, void* workspace, size_t ws_bytes, void* stream);

/* ---- GEMM with fused epilogue --------------------------------------------------------------
 * C[M,N] = alpha*opA(A)*opB(B) + beta*C + bias[N], then optional tanh.  Row-major; opA(A) is MxK:
 * transA=0 -> A stored [M,K] (lda>=K), transA=1 -> A stored [K,M] (lda>=M); opB(B) is KxN:
 * transB=0 -> B stored [K,N], transB=1 -> B stored [N,K] (an nn.Linear weight).  `batch` strided
 * instances (strides in elements).  Replaces nn.Linear at src/asr.py:307,316 (proj+tanh), :46,69
 * (ctc_layer), :384,419 (psi), :41,92 (char_trans), the x*W_ih^T half of nn.LSTM (:473-481) and
 * LSTMCell (:329-331), and their autograd backward (dX = dY*W, dW += dY^T*X). */
int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
             int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc,
             int64_t strideC, const float* bias, int act, int batch, void* stream);
/* out[n] = beta*out[n] + sum_m X[m,n]  (bias gradients of the layers above). */
int las_colsum(const float* X, int64_t ld, int M, int N, float beta, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
