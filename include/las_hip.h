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

/* ---- Persistent (Bi)LSTM recurrence, time-major -------------------------------------------------
 * Replaces the sequential half of nn.LSTM(bidirectional, batch_first, packed) and its BPTT at
 * src/asr.py:473-484, incl. pack/pad semantics (:480,483) and the drop/concat down-sampling (:487-497).
 * The input projection x*W_ih^T is a separate las_gemm.  Shapes (ND = 1|2 directions, gate order i,f,g,o):
 *   xproj [T][B][ND*4H] (no bias), b_ih/b_hh [ND*4H], w_hh [ND][4H][H], lens [B] (desc. order not required)
 *   y     [T_out][B][F_out] layer output in next-layer layout (see las_lstm_out_shape); zero at t>=len
 *   hf    [T][B][ND*H] hidden history (fp32); may alias y when sr==1
 *   hx    [ND][T][B][H] exchange copy in the MFMA operand type (bf16: 2 B/elem, f32: 4 B/elem)
 *   gates [T][B][ND*4H] post-activation gates, cs [T][B][ND*H] cell states (saved for bwd)
 *   sync  las_lstm_sync_bytes() bytes of scratch; status: int32, caller-zeroed, set to LAS_E_TIMEOUT if
 *         the in-kernel hand-off spin gave up (all workgroups then exit; outputs are garbage).
 * bwd: dy [T_out][B][F_out] -> dgf [T][B][ND*4H] (= d loss / d xproj, fp32) and dgx [ND][T][B][4H]
 * (exchange copy, operand type).  dW_ih, dW_hh, db, dx follow from dgf through las_gemm / las_colsum.
 * Limits: H % 8 == 0, B <= 128, ND*ceil(H/16) <= 256 (one workgroup per CU, all co-resident). */
void las_lstm_out_shape(int T, int H, int ND, int sr, int concat, int* T_out, int* F_out);
size_t las_lstm_sync_bytes(void);
int las_lstm_rec_fwd(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                     const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y, float* hf,
                     void* hx, float* gates, float* cs, void* sync, int* status, void* stream);
int las_lstm_rec_bwd(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                     const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx, float* dgf,
                     void* sync, int* status, void* stream);

/* ---- small data-movement / elementwise kernels ------------------------------------------------- */
/* out[d1][d0][:] = in[d0][d1][:] (batch-major <-> time-major; replaces the implicit layout of batch_first) */
int las_transpose01(const float* in, float* out, int D0, int D1, int F, void* stream);
/* out = dy*(1-y^2): backward of torch.tanh at src/asr.py:316,419 */
int las_tanh_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream);
/* lens[b] = #frames of x[b] whose feature sum != 0 (src/solver.py:134, on the host there) */
int las_infer_lengths(const float* x, int B, int T, int D, int32_t* lens, void* stream);
/* out[b] = #nonzero entries of y[b,:] (src/solver.py:136,159) */
int las_count_nonzero_i64(const int64_t* y, int B, int L, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
