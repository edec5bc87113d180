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
enum { LAS_ACT_NONE = 0, LAS_ACT_TANH = 1, LAS_ACT_RELU = 2 };

int las_abi_version(void);
const char* las_error_string(int code);

/* Start-up self-test of the tagged-granule hand-offs (csrc/selftest.hip): workgroup pairs play `iters` rounds of ping-pong through
 * 16-byte granules {hash(tag), tag} with the product kernels' store / poll instructions, inside one XCD (plain stores) and across
 * XCDs (sc1 stores); every polled value must be consistent with its own tag.  result (HOST memory, 3 words): inconsistent
 * observations, lanes that finished (16384 expected: 64 pairs x 2 sides x 64 lanes x 2 flavours), timeouts.  workspace: las_granule_selftest_bytes() device bytes.  Synchronises
 * the stream.  A caller that sees result[0] != 0 or result[1] != 16384 must not use the granule kernels (LAS_LSTM_NO_GR, LAS_DEC_NO_PK). */
size_t las_granule_selftest_bytes(void);
int las_granule_selftest(int iters, void* workspace, unsigned* result, void* stream);

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
 * C[M,N] = alpha*opA(A)*opB(B) + beta*C + bias[N], then optional tanh / ReLU.  Row-major; opA(A) is MxK:
 * transA=0 -> A stored [M,K] (lda>=K), transA=1 -> A stored [K,M] (lda>=M); opB(B) is KxN:
 * transB=0 -> B stored [K,N], transB=1 -> B stored [N,K] (an nn.Linear weight).  `batch` strided
 * instances (strides in elements).  Replaces nn.Linear at src/asr.py:307,316 (proj+tanh), :46,69
 * (ctc_layer), :384,419 (psi), :41,92 (char_trans), the x*W_ih^T half of nn.LSTM (:473-481) and
 * LSTMCell (:329-331), and their autograd backward (dX = dY*W, dW += dY^T*X). */
int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
             int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc,
             int64_t strideC, const float* bias, int act, int batch, void* stream);
/* The same product with bf16 SOURCE operands and / or a bf16 copy of the result (bf16 mode): a_bf16 / b_bf16 != 0 -> that
 * operand is a bf16 matrix in the same layout (lda / ldb / strides in elements; 16-byte aligned, ld % 8 == 0) -- the bf16
 * twins activations' producers write and the bf16 shadow of the weights (las_adam_step / las_adadelta_step); C16 != NULL ->
 * also C16[m*ldc16 + n] = bf16(C[m][n]).  Half the L2->LDS bytes of fp32 sources and no conversion while staging.
 * LAS_E_UNSUPPORTED if an operand is not aligned for it (the caller then passes its fp32 copy to las_gemm). */
int las_gemm_ex(int prec, int transA, int transB, int M, int N, int K, float alpha, const void* A, int a_bf16, int64_t lda,
                int64_t strideA, const void* B, int b_bf16, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc,
                int64_t strideC, const float* bias, int act, int batch, void* C16, int64_t ldc16, void* stream);
/* out[n] = beta*out[n] + sum_m X[m,n]  (bias gradients of the layers above). */
int las_colsum(const float* X, int64_t ld, int M, int N, float beta, float* out, void* stream);
/* The same sums added to TWO vectors in one pass over X (out2 may be NULL): nn.LSTM's / nn.LSTMCell's bias_ih and bias_hh
 * receive the same gradient (src/asr.py:473, :330). */
int las_colsum2(const float* X, int64_t ld, int M, int N, float beta, float* out, float* out2, void* stream);

/* ---- Persistent (Bi)LSTM recurrence, time-major -------------------------------------------------
 * Replaces the sequential half of nn.LSTM(bidirectional, batch_first, packed) and its BPTT at
 * src/asr.py:473-484, incl. pack/pad semantics (:480,483) and the drop/concat down-sampling (:487-497).
 * The input projection x*W_ih^T is a separate las_gemm.  Shapes (ND = 1|2 directions, gate order i,f,g,o):
 *   xproj [T][B][ND*4H] (no bias), b_ih/b_hh [ND*4H], w_hh [ND][4H][H], lens [B] (desc. order not required)
 *   y     [T_out][B][F_out] layer output in next-layer layout (see las_lstm_out_shape); zero at t>=len
 *   hf    [T][B][ND*H] hidden history (fp32); may alias y when sr==1
 *   hx    exchange workspace of las_lstm_hx_bytes() bytes: a ring of the last four steps' h, either [ND][4][B][Hx] in the
 *         MFMA operand type (bf16: Hx = H rounded up to 8; f32: rounded up to 4) or, in bf16 mode, tagged 16-byte
 *         granules {6 batch rows of one unit, step tag} (zeroed by the call); when Hx != H the caller zero-fills it once
 *   gates [T][B][ND*4H] post-activation gates, cs [T][B][ND*H] cell states (saved for bwd)
 *   sync  las_lstm_sync_bytes() bytes of scratch; status: int32, caller-zeroed, set to LAS_E_TIMEOUT if
 *         the in-kernel hand-off spin gave up (all workgroups then exit; outputs are garbage).
 * bwd: dy [T_out][B][F_out] -> dgf [T][B][ND*4H] (= d loss / d xproj, fp32); dgx is the exchange workspace of
 * las_lstm_bwd_ws_bytes() bytes (a ring of partial-dh inboxes, or the dgates copy of the all-gather variant).  dW_ih, dW_hh, db, dx follow from dgf through las_gemm / las_colsum.
 * Limits: H % 2 == 0, B <= 2048 (the batch is cut into independent slices of <= 128 rows, normally ~12), ND*ceil(H/16) <= 256 (one workgroup per CU, all co-resident). */
void las_lstm_out_shape(int T, int H, int ND, int sr, int concat, int* T_out, int* F_out);
size_t las_lstm_sync_bytes(void);
size_t las_lstm_hx_bytes(int prec, int T, int B, int H, int ND);        /* size of the `hx` exchange workspace of rec_fwd */
size_t las_lstm_bwd_ws_bytes(int prec, int T, int B, int H, int ND);   /* size of the `dgx` exchange workspace of rec_bwd */
int las_lstm_bwd_is_ksplit(int prec, int T, int B, int H, int ND);     /* which kernel rec_bwd runs: 3 lstm_bwd_x32_kernel (512 < H <= 1024: 32 units per workgroup, every group inside one XCD), 2 lstm_bwd_gr_kernel (reduce-scatter of partial dh in tagged granules), 1 lstm_bwd_ks_kernel (the same behind a flag), 0 lstm_bwd_kernel (all-gather of dgates) */
int las_lstm_fwd_variant(int prec, int T, int B, int H, int ND);       /* which kernel rec_fwd runs: 2 lstm_fwd_x32_kernel (512 < H <= 1024, 32 units per workgroup), 1 lstm_fwd_gr_kernel (tagged granules), 0 lstm_fwd_kernel */
int las_lstm_resident_wgs(int prec, int T, int B, int H, int ND);      /* workgroups rec_fwd / rec_bwd keep resident for the whole launch (one per CU, all must be co-resident); callers that overlap other device work with the recurrence (RCCL collectives: dist.py) use it to decide whether that work finds free CUs */
/* y_bf16 / dgf_bf16 (may be null): bf16 twins of y [T_out][B][F_out] / dgf [T][B][ND*4H], the operands of the GEMMs behind the
 * launch (projection; d x, d W_ih, d W_hh).  The granule and 32-unit kernels write them next to the fp32 stores; behind the
 * other kernels the call appends one cast pass. */
int las_lstm_rec_fwd(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                     const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y, float* hf,
                     void* y_bf16, void* hx, float* gates, float* cs, void* sync, int* status, void* stream);
/* rec_fwd with the layer's input projection fused (x [T][B][I], w_ih [ND*4H][I] instead of xproj): bf16 mode, granule kernel, one
 * batch tile per slice, I <= 96 and a multiple of 4 -- las_lstm_fwd_fx_ok says whether a shape qualifies (else LAS_E_UNSUPPORTED and
 * the caller runs las_gemm + las_lstm_rec_fwd).  The bottom layer's projection is K = I = 80: 295 MB of output at the C2 shape. */
int las_lstm_fwd_fx_ok(int prec, int T, int B, int H, int ND, int I);
int las_lstm_rec_fwd_fx(int prec, const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                        const float* w_hh, const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y, float* hf,
                        void* y_bf16, void* hx, float* gates, float* cs, void* sync, int* status, void* stream);
int las_lstm_rec_bwd(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                     const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx, float* dgf,
                     void* dgf_bf16, void* sync, int* status, void* stream);

/* ---- small data-movement / elementwise kernels ------------------------------------------------- */
/* out[d1][d0][:] = in[d0][d1][:] (batch-major <-> time-major; replaces the implicit layout of batch_first) */
int las_transpose01(const float* in, float* out, int D0, int D1, int F, void* stream);
/* out = dy*(1-y^2): backward of torch.tanh at src/asr.py:316,419 */
int las_tanh_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream);
/* the same, also writing the bf16 twin of `out` (operand of the dX / dW GEMMs that follow);  out_bf16 = bf16(in) */
int las_tanh_bwd_twin(const float* dy, const float* y, float* out, void* out_bf16, int64_t n, void* stream);
int las_cast_bf16(const float* in, void* out_bf16, int64_t n, void* stream);
/* lens[b] = #frames of x[b] whose feature sum != 0 (src/solver.py:134, on the host there) */
int las_infer_lengths(const float* x, int B, int T, int D, int32_t* lens, void* stream);
/* out[b] = #nonzero entries of y[b,:] (src/solver.py:136,159) */
int las_count_nonzero_i64(const int64_t* y, int B, int L, int32_t* out, void* stream);

/* ---- per-step skinny products (M = batch <= 128) --------------------------------------------------
 * nn.LSTMCell (src/asr.py:329-331, called :353-355): gates = x W_ih^T + b_ih + h W_hh^T + b_hh, then
 * c' = f*c + i*g, h' = o*tanh(c').  gates_out [B][4C] receives the post-activation gates (i,f,g,o). */
int las_lstm_cell_fwd(int prec, const float* x, int64_t ldx, int Kx, const float* h_prev, const float* c_prev,
                      const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int B, int C,
                      float* h_out, float* c_out, float* gates_out, void* stream);
/* out[b][n] (+)= act(sum_k x[b][k] w[n][k] + bias[n]); act: 0 none, 1 tanh.  (phi at src/asr.py:383,422;
 * per-step char_trans :92; backward products with transposed weight copies.) */
int las_skinny_linear(int prec, const float* x, int64_t ldx, const float* w, int64_t ldw, int B, int N, int K,
                      const float* bias, int act, int accumulate, float* out, int64_t ldo, void* stream);

/* ---- attend-and-spell decoder loop ------------------------------------------------------------------
 * One call runs all L decode steps of Seq2Seq.forward (src/asr.py:77-107): Attention.forward (:410-457,
 * 'dot' / 'loc'), the Speller's LSTMCell stack (:352-357), embedding look-ups and the teacher / sampled /
 * greedy feedback (:95-102).  All pointers below are device pointers; the structs themselves are host. */
typedef struct {
    int B, Tp, E, A, C, NL, V, L;   /* batch, encoder frames, enc dim, att dim, dec dim, dec layers, vocab, steps */
    int loc;                        /* 0: dot attention, 1: location-aware */
    int prec;
    float dropout;                  /* Speller dropout (asr.py:327,353,355): on the cell-0 input and on the recurrent
                                       state of layers >= 1; 0 disables (eval mode).  Masks are a counter hash of
                                       (drop_seed, step, layer, element): las_dropout_rows with las_decoder_drop_seed() */
    unsigned drop_seed;
} las_dec_dims;
typedef struct {
    const float* emb;               /* embed.weight [V][C] */
    const float* w_phi;             /* attention.phi.weight [A][C] */
    const float* conv_w;            /* attention.loc_conv.weight [10][1][201]   (loc) */
    const float* w_lp;              /* attention.loc_proj.weight [A][10]        (loc) */
    const float* w_e;               /* attention.gen_energy.weight [1][A]       (loc) */
    const float* b_e;               /* attention.gen_energy.bias [1]            (loc) */
    const float* w_ih[4];           /* decoder.layer{l}.weight_ih [4C][C+E | C] */
    const float* w_hh[4];           /* decoder.layer{l}.weight_hh [4C][C] */
    const float* b_ih[4];
    const float* b_hh[4];
    const float* w_char;            /* char_trans.weight [V][C] (only read for sampled / greedy steps) */
    const float* b_char;
    /* transposed copies for the backward products (k-contiguous rows), refreshed by the caller */
    const float* w_ihT[4];          /* [C+E | C][4C] */
    const float* w_hhT[4];          /* [C][4C] */
    const float* w_phiT;            /* [C][A] */
    /* optional (bf16 mode): the weight operand of each M = batch product pre-packed in MFMA fragment order by
       las_skinny_pack_weights (one coalesced 1 KB read per wave and k-step); NULL: read the matrices above */
    const void* pk_phi;             /* N = A: w_phi */
    const void* pk_cell[4];         /* cell l: N = 4C (cell_mode), segments w_ih[l], w_hh[l] */
    const void* pk_dx[4];           /* N = C+E | C: w_ihT[l] */
    const void* pk_dh[4];           /* N = C: w_hhT[l] (l = 0: segments w_hhT[0], w_phiT) */
} las_dec_params;
typedef struct {                    /* saved activations, written by fwd, read by bwd (caller-owned) */
    int32_t* tok;                   /* [L][B] token fed at each step */
    float* xin;                     /* [L][B][C+E] cell-0 input (embedding | context) */
    float* q;                       /* [L][B][A] tanh(phi h0_{t-1}) */
    float* att;                     /* [L+1][B][Tp] slot t+1 = attention of step t; slot 0 = initial prev_att */
    float* hs;                      /* [NL][L+1][B][C] slot t+1 = h_t; slot 0 = 0 */
    float* cs;                      /* [NL][L+1][B][C] */
    float* gates;                   /* [NL][L][B][4C] post-activation */
    float* f;                       /* loc: [L][B][10][Tp] location features */
    void* s;                        /* loc: [L][B][Tp][A] tanh(psi + q + u): fp32 in LAS_PREC_F32; in LAS_PREC_BF16 a 16-bit code per
                                     * element (bf16 of copysign(1 - |s|, s): las_common.h), las_decoder_s_elem_bytes() bytes each */
    float* ebuf;                    /* [B][Tp] scratch */
    float* logits_step;             /* [B][V] scratch (sampled / greedy steps) */
    float* xdrop;                   /* dropout > 0: [L][B][C+E] cell-0 input after dropout (xin keeps the clean one) */
    float* hdrop;                   /* dropout > 0, NL > 1: [NL][L][B][C] recurrent state of layers >= 1 after dropout */
    void* pk_ws;                    /* las_decoder_pk_workspace_bytes(dims) bytes, or NULL: with it, a teacher-forced loc-attention
                                       loop with one Speller layer and no dropout runs as ONE persistent launch (decoder_pk.hip) */
    int32_t* pk_status;             /* int32, caller-zeroed: set to LAS_E_TIMEOUT if a hand-off spin of that launch ran out */
} las_dec_state;
/* 0: this shape / mode always takes the per-step launch path */
size_t las_decoder_pk_workspace_bytes(const las_dec_dims* dims);
/* step_mode[t] (host): how the token fed at step t is chosen: 1 teacher y[:,t], 0 sampled from softmax of
 * step t-1's logits, 2 argmax of step t-1's logits.  step_mode==NULL means all teacher.  y [B][Ly] int64. */
int las_decoder_fwd(const las_dec_dims* dims, const las_dec_params* params, const float* enc, const float* psi,
                    const int32_t* enc_len, const int64_t* y, int Ly, const uint8_t* step_mode, unsigned seed,
                    las_dec_state* state, void* stream);

/* tok[r] = a draw from softmax(logits[r][:]) (greedy == 0; counter-hash Gumbel-max: same distribution as
 * `Categorical(F.softmax(cur_char)).sample()`, src/asr.py:99, not the same stream) or argmax (greedy != 0, src/asr.py:102):
 * the token choice las_decoder_fwd makes between two steps, with seed = its seed + 0x9e3779b9 * step. */
int las_sample_rows(const float* logits, int rows, int V, int greedy, unsigned seed, int32_t* tok, void* stream);

typedef struct {                    /* backward buffers (caller-owned); the driver zeroes what it accumulates into */
    float* dgates;                  /* [NL][L][B][4C]  d loss / d gate pre-activations */
    float* dxin;                    /* [L][B][C+E]     d loss / d cell-0 input (embedding | context) */
    float* dq_pre;                  /* [L][B][A]       d loss / d (phi h) before the tanh */
    float* de;                      /* [L][B][Tp] d loss / d energy */
    float* dh_carry;                /* [NL][B][C] */
    float* dc_carry;                /* [NL][B][C] */
    float* d_below;                 /* [B][C] scratch */
    float* da;                      /* [B][Tp] scratch */
    float* df;                      /* loc: [L][B][10][Tp] d loss / d location features of each step (carries the
                                       gradient through the location conv to the previous step's attention) */
    float* dpsi;                    /* loc: [B][Tp][A] d loss / d psi(enc), summed over the steps after the loop */
    float* acc;                     /* loc: [B][las_decoder_loc_acc_floats(A)] per-utterance partial sums:
                                       d w_lp^T [10][A] | d w_e [A] | d b_e [1] | pad to 4 | d conv_w [10*201] */
    float* demb;                    /* [V][C] d loss / d embed.weight */
    void* pk_ws;                    /* las_decoder_pk_bwd_workspace_bytes(dims) bytes, or NULL: with it (and the conditions of the
                                       forward's pk_ws) the whole BPTT chain runs as ONE persistent launch (decoder_pk_bwd.hip) */
    int32_t* pk_status;             /* int32, caller-zeroed: LAS_E_TIMEOUT if a hand-off spin of that launch ran out */
    const void* enc_bf16;           /* optional: enc as bf16 [B][Tp][E] (the twin its producer made for the psi GEMM).  bf16 mode, per-step
                                       chain: d a = enc . d ctx reads it instead of the fp32 rows (half the bytes and row registers) */
} las_dec_bwd_state;
size_t las_decoder_pk_bwd_workspace_bytes(const las_dec_dims* dims);
int64_t las_decoder_loc_acc_floats(int A);
int las_decoder_att_chunks(int Tp);
size_t las_decoder_s_elem_bytes(int prec);          /* element size of las_dec_state.s: 4 (fp32) | 2 (LAS_PREC_BF16: the 16-bit code) */
/* g_htop [L][B][C]: gradient wrt the top-layer hidden state of every step (from char_trans).  After this call
 * the remaining sums over steps are plain contractions for las_gemm / las_colsum:
 *   dW_ih[l] = dgates[l]^T x_l, dW_hh[l] = dgates[l]^T hs[l][0:L], db = colsum(dgates[l]), dW_phi = dq_pre^T hs[0][0:L],
 *   d enc[b] = att[1:,b]^T dxin[:,b,C:],  dot: d psi[b] = de[:,b]^T q[:,b]. */
int las_decoder_bwd(const las_dec_dims* dims, const las_dec_params* params, const float* enc, const float* psi,
                    const int32_t* enc_len, const las_dec_state* state, const float* g_htop,
                    las_dec_bwd_state* bwd, void* stream);
/* The same call in two parts, for a caller that keeps the parameter-only sums off its critical path (a second stream):
 *   LAS_DEC_BWD_CHAIN       the zero fills, the BPTT chain and att_loc_post (d psi, d w_lp, d w_e, d b_e into acc): everything
 *                           the encoder's backward waits for;
 *   LAS_DEC_BWD_PARAM_SUMS  d conv_w (into acc, behind the chain's zero fill) and the embedding rows demb: parameters only.
 * PARAM_SUMS reads what CHAIN wrote (df, dxin): enqueue it behind CHAIN.  las_decoder_bwd = both, in this order. */
#define LAS_DEC_BWD_CHAIN 1
#define LAS_DEC_BWD_PARAM_SUMS 2
int las_decoder_bwd_parts(const las_dec_dims* dims, const las_dec_params* params, const float* enc, const float* psi,
                          const int32_t* enc_len, const las_dec_state* state, const float* g_htop, las_dec_bwd_state* bw,
                          int parts, void* stream);

/* out[c][r] = in[r][c] (weight transposes for the backward skinny products) */
int las_transpose2d(const float* in, float* out, int R, int C, void* stream);

/* ---- joint loss -----------------------------------------------------------------------------------------
 * Attention CE: CrossEntropyLoss(ignore_index=0,'none') on att_pred [B][L][V] against label = y[:,1:L+1],
 * sum_t / ntok_b (ntok_b = #(y[b]!=0)), batch mean (src/solver.py:90,149-155).  One pass also writes
 * dlogits = gscale * d loss / d logits (pass NULL to skip).  rowloss [B*L] scratch, loss [1]. */
int las_ce_loss(const float* logits, const int64_t* y, int Ly, const int32_t* ntok, int B, int L, int V, float gscale,
                float* rowloss, float* loss, float* dlogits, void* stream);
/* CTCLoss reduction='mean': out[0] = mean_b x[b]/max(n[b],1) (src/solver.py:93,160) and its gradient
 * gx[b] = g[0]*scale/(B*max(n[b],1)). */
int las_norm_mean_fwd(const float* x, const int32_t* n, int B, float* out, void* stream);
int las_norm_mean_bwd(const float* g, float scale, const int32_t* n, int B, float* gx, void* stream);
/* x *= alpha[0] (device scalar);  out[0] = wa*a[0] + wb*b[0]  ((1-w)*att + w*ctc, src/solver.py:163) */
int las_scale_dev(float* x, int64_t n, const float* alpha, void* stream);
int las_combine2(const float* a, float wa, const float* b, float wb, float* out, void* stream);

/* ---- clip + optimiser over one flat fp32 vector ------------------------------------------------------
 * las_grad_norm: out3 = { ||gscale*g||_2, clip coefficient min(1, max_norm/(norm+1e-6))*gscale, skip flag (1 if the
 * norm is NaN) } and, unless skipped, step_dev[0] += 1 -- torch.nn.utils.clip_grad_norm_(params, 5) + the
 * math.isnan guard at src/solver.py:178-181, without the host sync.  gscale = 1/world_size after a sum all-reduce.
 * las_adam_step / las_adadelta_step: torch.optim.Adam(lr, betas, eps=1e-8) / Adadelta(lr, rho=0.9, eps=1e-8)
 * (src/solver.py:101-106,182) applied to coef*g; optionally zero g afterwards (opt.zero_grad, solver.py:139). */
size_t las_grad_norm_workspace_bytes(void);
int las_grad_norm(const float* g, int64_t n, float gscale, float max_norm, void* workspace, float* out3,
                  int32_t* step_dev, void* stream);
/* p_bf16 (optional, same length as p): receives bf16(p) after the update -- the weights' bf16 shadow, operand of las_gemm_ex */
int las_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                  const float* norm3, const int32_t* step_dev, int zero_grad, void* p_bf16, void* stream);
int las_adadelta_step(float* p, float* g, float* sq, float* acc, int64_t n, float lr, float rho, float eps,
                      const float* norm3, int zero_grad, void* p_bf16, void* stream);

/* Weight operand of a skinny product (las_skinny_linear / las_lstm_cell_fwd / the decoder's per-step products) packed as
 * bf16 MFMA fragments: for every block of 16 output rows and every 32-wide k-step over the concatenated segments,
 * 64 lanes x 8 values; cell_mode: rows are the gate-interleaved rows of las_lstm_cell_fwd (N = 4C). */
size_t las_skinny_pack_bytes(int N, int K0, int K1, int K2, int cell_mode, int C);
int las_skinny_pack_weights(const float* w0, int64_t ldw0, int K0, const float* w1, int64_t ldw1, int K1,
                            const float* w2, int64_t ldw2, int K2, int N, int cell_mode, int C, void* out, void* stream);

/* inverted dropout on rows: out[r][i] = keep ? in[r][i] / (1-p) : 0, keep = hash(seed, r*N+i) >= p (nn.Dropout at
 * asr.py:327; same distribution, not torch's stream).  in == out allowed.  las_decoder_drop_seed: the seed the decoder
 * uses for (step, layer). */
int las_dropout_rows(const float* in, int64_t ld_in, float* out, int64_t ld_out, int R, int N, float p, unsigned seed,
                     void* stream);
unsigned las_decoder_drop_seed(unsigned drop_seed, int step, int layer);

/* training-time accuracy on the device: pred = argmax rows of att_pred; token accuracy up to the first 0 label,
 * mean over utterances (np.argmax + cal_acc at src/postprocess.py:121-133, called every step at solver.py:187) */
int las_argmax_rows(const float* logits, int rows, int V, int32_t* pred, void* stream);
int las_token_acc(const int32_t* pred, const int64_t* y, int Ly, int B, int L, float* out, void* stream);

/* ---- VGG front-end of the Listener -----------------------------------------------------------
 * Replaces VGGExtractor.forward (src/asr.py:546-558; view_input :533-544, check_dim :522-531) and its autograd
 * backward: x [B][T][D] (D = C_in*F, delta channel outermost; F = 13 if D%13==0 else 40 if D%40==0) ->
 * conv3x3(C_in->64)+ReLU, conv3x3(64->64)+ReLU, MaxPool2d(2), conv3x3(64->128)+ReLU, conv3x3(128->128)+ReLU,
 * MaxPool2d(2) -> out [B][T/4][128*(F/4)] (time_major=0) or [T/4][B][128*(F/4)] (time_major=1), feature index
 * c*(F/4)+f as the reference's transpose+view.  The T%4 tail frames are dropped; lengths are the caller's (//4).
 * Activations are kept channels-last [B][T][F][C]; all buffers are caller-owned (sizes from las_vgg_get_dims):
 *   y1,y2 [R1][64]  p1 [R2][64]  y3,y4 [R2][128]  idx1 [R2][64] bytes  idx2 [R3][128] bytes
 *   col [col_floats] scratch (zero-padded 3x3 patch matrix of conv1, the only one materialised)
 *   wr [wr_floats] weights re-ordered to [C_out][tap][C_in], then conv2..4 again as [C_in][8-tap][C_out] for the
 *      data gradient (written by fwd, read by bwd)
 *   dwr [wr_floats], ga, gb [R1][64]   (bwd only)
 * w[i] [C_out][C_in][3][3] and b[i] [C_out] are the reference's conv{1..4}.weight/bias; bwd ACCUMULATES into
 * dw[i]/db[i] and, when dx != NULL, writes dx [B][T][D] (zero in the dropped tail). */
typedef struct {
    int C_in, F, Tt, T2, F2, T4, F4, out_dim;   /* Tt = T - T%4, T2 = Tt/2, F2 = F/2, T4 = T2/2, F4 = F2/2 */
    int Kp[4];                                   /* patch-row length of conv i: 9*C_in(i) rounded up to 32 */
    int64_t R1, R2, R3;                          /* B*Tt*F, B*T2*F2, B*T4*F4 */
    int64_t col_floats, wr_floats;
} las_vgg_dims;
typedef struct { const float* w[4]; const float* b[4]; } las_vgg_params;
typedef struct { float* dw[4]; float* db[4]; } las_vgg_grads;
typedef struct {
    float *y1, *y2, *p1, *y3, *y4;
    uint8_t *idx1, *idx2;
    float *col, *wr, *dwr, *ga, *gb;
} las_vgg_state;
int las_vgg_get_dims(int B, int T, int D, las_vgg_dims* dims);
int las_vgg_fwd(int prec, const float* x, int B, int T, int D, const las_vgg_params* params, las_vgg_state* state,
                float* out, int time_major, void* stream);
int las_vgg_bwd(int prec, const float* x, const float* dout, int B, int T, int D, int time_major,
                const las_vgg_state* state, const las_vgg_grads* grads, float* dx, void* stream);

/* ---- joint CTC/attention beam search (decode path) ----------------------------------------------
 * One decode step for B hypotheses that resume from caller-provided states (Seq2Seq.beam_decode's inner loop,
 * src/asr.py:205-215): dims->L must be 1; state slabs are those of las_decoder_fwd with L = 1, of which the
 * caller fills slot 0 of hs/cs ([NL][2][B][C]), att[0] ([B][Tp], previous attention; loc only) and tok[0..B)
 * (token fed at this step).  Writes slot 1 of hs/cs, att[1], xin, q, gates and logits [B][V] = char_trans(h_top)
 * (asr.py:214).  enc/psi are per-hypothesis ([B][Tp][.]): replicate the utterance's encoding B times. */
int las_decoder_step(const las_dec_dims* dims, const las_dec_params* params, const float* enc, const float* psi,
                     const int32_t* enc_len, las_dec_state* state, float* logits, void* stream);
/* out[r][:] = log_softmax(x[r][:])  (F.log_softmax at asr.py:181,215) */
int las_log_softmax_rows(const float* x, int R, int V, float* out, void* stream);
/* k largest entries of every row, descending, ties to the lower index (tensor.topk at asr.py:219,237); V*4 B <= 60 KB */
int las_topk_rows(const float* x, int R, int V, int k, float* vals, int32_t* idx, void* stream);
/* CTCPrefixScore (src/ctc.py): lp [T][V] log-probs of one utterance (blank = 0).  init: r0 [T][2] (ctc.py:19-27).
 * score = cheap_compute (ctc.py:65-101) for N hypotheses x K candidates at once: r_prev [N][T][2], last_tok /
 * prefix_len [N] (last_tok ignored when prefix_len = 0), cand [N][K] -> psi [N][K], r_out [N][K][T][2].
 * float32 arithmetic with numpy's logaddexp, and the reference's treatment of cand == last_tok (it drops the
 * blank-ending path of the prefix) kept as is. */
int las_ctc_prefix_init(const float* lp, int T, int V, float* r0, void* stream);
int las_ctc_prefix_score(const float* lp, int T, int V, const float* r_prev, const int32_t* last_tok,
                         const int32_t* prefix_len, const int32_t* cand, int N, int K, float* psi, float* r_out,
                         void* stream);
/* cur [N][V] (in place) = (1-w)*cur + w*hack, hack = -1e6 except hack[cand[n][j]] = psi[n][j] - prev_ctc[n];
 * then cur[n][0] = -1e7 (asr.py:218-229) */
int las_beam_combine(float* cur, int N, int V, const int32_t* cand, const float* psi, const float* prev_ctc, int K,
                     float ctc_weight, void* stream);

#ifdef __cplusplus
}
#endif
#endif
