/* CPU oracle: CTC forward-backward with fused log-softmax  --  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of what the reference obtains from
 *   torch.nn.CTCLoss(blank=0, reduction='mean')(F.log_softmax(ctc_pred.transpose(0,1), -1),
 *                                               label, LongTensor(enc_len), target_len)
 * (/root/reference/src/solver.py:93,160).  The arithmetic lives in third-party PyTorch
 * (ATen ctc_loss_cpu / ctc_loss_backward_cpu, not vendored in the reference); this file restates the
 * published algorithm (Graves et al. 2006, log-space alpha/beta over the blank-extended label) and is
 * pinned by tests/golden/g2_ctc_*.npz, which were produced by that ATen code via tools/gen_golden.py.
 *
 * Layouts: logits [B][T][V] fp32 (raw, batch-major as asr.py:69 emits them); label [B][L] int32
 * zero-padded; log_alpha [B][T][2L+1] (-inf outside the lattice); grad [B][T][V] = d(sum_b nll_b)/dlogits.
 * Infeasible utterances (nll=+inf) get NaN gradients, as ATen does with zero_infinity=False.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

int ctc_ref(const float* logits, const int32_t* label, const int32_t* enc_len, const int32_t* tgt_len,
            int B, int T, int V, int L, int blank,
            float* nll, float* log_alpha, float* grad /* may be NULL */) {
    const int S = 2 * L + 1;
    double* lp = (double*)malloc(sizeof(double) * (size_t)T * V);
    double* al = (double*)malloc(sizeof(double) * (size_t)T * S);
    double* be = (double*)malloc(sizeof(double) * (size_t)T * S);
    int* ext = (int*)malloc(sizeof(int) * S);
    if (!lp || !al || !be || !ext) return -1;
    for (int b = 0; b < B; ++b) {
        const int Tb = enc_len[b], n = tgt_len[b], Sn = 2 * n + 1;
        const float* x = logits + (size_t)b * T * V;
        for (int t = 0; t < T; ++t) {                       /* log-softmax rows */
            double m = -INFINITY, s = 0;
            for (int v = 0; v < V; ++v) if (x[(size_t)t * V + v] > m) m = x[(size_t)t * V + v];
            for (int v = 0; v < V; ++v) s += exp(x[(size_t)t * V + v] - m);
            for (int v = 0; v < V; ++v) lp[(size_t)t * V + v] = x[(size_t)t * V + v] - m - log(s);
        }
        for (int s = 0; s < Sn; ++s) ext[s] = (s & 1) ? label[(size_t)b * L + s / 2] : blank;
        for (int i = 0; i < T * S; ++i) { al[i] = -INFINITY; be[i] = -INFINITY; }
        al[0] = lp[blank];
        if (Sn > 1) al[1] = lp[ext[1]];
        for (int t = 1; t < Tb; ++t)
            for (int s = 0; s < Sn; ++s) {
                double a = al[(t - 1) * S + s];
                if (s > 0) a = lse2(a, al[(t - 1) * S + s - 1]);
                if (s > 1 && ext[s] != blank && ext[s] != ext[s - 2]) a = lse2(a, al[(t - 1) * S + s - 2]);
                al[t * S + s] = a + lp[(size_t)t * V + ext[s]];
            }
        double ll = al[(Tb - 1) * S + Sn - 1];
        if (Sn > 1) ll = lse2(ll, al[(Tb - 1) * S + Sn - 2]);
        nll[b] = (float)(-ll);
        for (int i = 0; i < T * S; ++i) log_alpha[(size_t)b * T * S + i] = (float)al[i];
        if (!grad) continue;
        float* g = grad + (size_t)b * T * V;
        for (size_t i = 0; i < (size_t)T * V; ++i) g[i] = 0.f;
        if (ll == -INFINITY) {
            for (size_t i = 0; i < (size_t)Tb * V; ++i) g[i] = NAN;
            continue;
        }
        be[(Tb - 1) * S + Sn - 1] = lp[(size_t)(Tb - 1) * V + blank];
        if (Sn > 1) be[(Tb - 1) * S + Sn - 2] = lp[(size_t)(Tb - 1) * V + ext[Sn - 2]];
        for (int t = Tb - 2; t >= 0; --t)
            for (int s = 0; s < Sn; ++s) {
                double a = be[(t + 1) * S + s];
                if (s + 1 < Sn) a = lse2(a, be[(t + 1) * S + s + 1]);
                if (s + 2 < Sn && ext[s + 2] != blank && ext[s + 2] != ext[s]) a = lse2(a, be[(t + 1) * S + s + 2]);
                be[t * S + s] = a + lp[(size_t)t * V + ext[s]];
            }
        for (int t = 0; t < Tb; ++t) {
            for (int v = 0; v < V; ++v) g[(size_t)t * V + v] = (float)exp(lp[(size_t)t * V + v]);
            for (int s = 0; s < Sn; ++s) {
                double ab = al[t * S + s] + be[t * S + s];
                if (ab > -INFINITY) g[(size_t)t * V + ext[s]] -= (float)exp(ab - ll - lp[(size_t)t * V + ext[s]]);
            }
        }
    }
    free(lp); free(al); free(be); free(ext);
    return 0;
}
