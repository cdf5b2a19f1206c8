// Internal declarations of the training-step kernels (train_kernels.hip, wgrad.hip, attention_bwd.hip).
#pragma once
#include "sr3_common.h"

namespace sr3 {

// GroupNorm(+SiLU) backward over the virtual concat (x0|x1): dA (grad w.r.t. the activated input) is
// overwritten with du, partial sums go to `part`, group sums to gs[B][G][2], parameter gradients to
// dgamma/dbeta[C], and dx0/dx1 (+=) receive the input gradient.  mr[B][G][2] = (mean, rstd).
int act_bwd(float* dA, const float* x0, const float* x1, int C0, int C1, int B, int HW, const float* ss, const float* mr,
            int groups, int act, const float* gamma, double* part, double* gs, float* dgamma, float* dbeta, float* dx0,
            float* dx1, hipStream_t st, unsigned drop_seed = 0, unsigned drop_thresh = 0, float drop_scale = 1.f,
            float* aout = nullptr, bool acc0 = true, bool acc1 = true);      // aout: also write a = dropout(act(x*scale+shift)) (what apply_act would, in the same pass; round 6)
// a = dropout(act(x*scale+shift)) over the virtual concat, materialised for the weight-gradient GEMM
int apply_act(const float* x0, const float* x1, int C0, int C1, int B, int HW, const float* ss, int act, unsigned drop_seed,
              unsigned drop_thresh, float drop_scale, float* out, hipStream_t st);
size_t act_bwd_part_bytes(int B, int HW, int C);
// acc0 / acc1 (here and in act_bwd): false = the first contribution to that destination in the backward walk is a plain store (round 6:
// the gradient mirror is no longer zeroed every step)
int grad_route(const float* g, int C0, int C1, int B, int Hs, int Ws, int ups, float* d0, float* d1, hipStream_t st,
               bool acc0 = true, bool acc1 = true);
int zero_insert(const float* g, int B, int Ho, int Wo, int C, float* z, hipStream_t st);
int w_flip_transpose(const float* w, int Cout, int taps, int Cin, int CoutP, float* wt, hipStream_t st);
int colsums(const float* g, int B, int HW, int C, double* part, float* dbias, float* dfilm, int film_stride,
            hipStream_t st);
int l1_loss_grad(const float* z, const float* e, int B, int Cc, int HW, int CP, float scale, bool l2, float* g_nhwc,
                 double* loss_part, float* loss_out, hipStream_t st);
int adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
              hipStream_t st);
int nchw_to_nhwc_pad(const float* a, int Ca, const float* b, int Cb, int B, int HW, int CP, float* out, hipStream_t st);

// Weight gradient of a conv (wgrad.hip): dw[n][tap][c] = sum_m dy[m][n] * a_tap[m][c], a = prologue(x0|x1)
// recomputed on the fly (same ConvParams geometry as the forward; p.out / p.w unused).
// `slabs` holds the per-pixel-split partial results; dw is overwritten.
struct WgradParams {
  ConvParams c;        // forward geometry + src0/src1/ss/act (+ ups/stride/ksize)
  const float* dy;     // [B,Ho,Wo,Cout]
  float* dw;           // [Cout][taps][Cin]
  float* slabs;        // [msplit][Cout][taps][Cin]
  int msplit;
};
size_t wgrad_slab_bytes(const ConvParams& c, int* msplit_out);
int conv_wgrad(const WgradParams& p, hipStream_t st);

// Attention backward (attention_bwd.hip): qkv [B][N][3C], dout [B][N][C] -> dqkv [B][N][3C] (overwritten)
// out_fwd (the forward attention output [B][N][C]) is only read by the key-blocked path (N too large for LDS strips)
// scratch (attention_backward_scratch_bytes): dK / dV slabs per query block, summed in order -- no atomics, no memset; null: fp32 atomics
size_t attention_backward_scratch_bytes(int B, int N, int C);
int attention_backward(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv, hipStream_t st,
                       float* scratch = nullptr, size_t scratch_bytes = 0);

// Embedding / FiLM backward (small): see train_small.hip
struct EmbedBwdParams {
  int variant, B, inner, F;
  const float* level; const int64_t* tstep; const float* freq;
  const float* w1; const float* b1; const float* w2; const float* b2; const float* wf;
  const float* dfilm;     // [B][F]
  float* dw1; float* db1; float* dw2; float* db2; float* dwf; float* dbf;
  float* scratch;         // >= B * 29 * inner floats (8 per image of the recompute, 5 of the MLP backward, 16 row-chunk partials)
};
int embed_backward(const EmbedBwdParams& p, hipStream_t st);

}  // namespace sr3
