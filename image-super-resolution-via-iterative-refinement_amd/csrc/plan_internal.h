// Internal plan structures shared by plan.hip (inference driver) and train_plan.hip (training step).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/sr3_mi355x.h"
#include "sr3_common.h"

namespace sr3 {

// ---------------------------------------------------------------------------------------------
// plan data
// ---------------------------------------------------------------------------------------------
struct Tensor {
  size_t off = 0;       // byte offset in the workspace
  size_t bytes = 0;
  int C = 0, H = 0, W = 0;
  size_t stat_off = 0;  // byte offset of the partial statistics [B][T][C][2] doubles (valid once stats_done)
  int stat_T = 0;       // partials per image
  bool stats_done = false;
  bool valid = false;
};

struct ResLayer {
  std::string name;     // e.g. "downs.1"
  int cin, cout, skip;  // cin includes skip
  bool attn;
  int film_off;         // row offset in the FiLM table
  size_t gn1_w, gn1_b, c1_w, c1_b, gn2_w, gn2_b, c2_w, c2_b, rc_w, rc_b;
  bool has_rc;
  size_t an_w, an_b, qkv_w, ao_w, ao_b;
};

struct Layer {
  int kind;  // 0 conv_in, 1 res, 2 down, 3 up
  std::string name;
  int cin, cout;
  size_t w, b;  // conv_in / down / up
  ResLayer res;
};

// numeric values are part of sr3_unet_forward_profile's op_kind encoding (kind * 10)
enum OpKind { OP_RESERVED, OP_EMBED, OP_CONV_IN, OP_STATS, OP_FOLD, OP_CONV, OP_ATTN, OP_CONV_OUT };

struct Op {
  OpKind kind;
  // generic offsets (bytes into workspace unless noted)
  size_t a = 0, b = 0, c = 0, d = 0, e = 0, f = 0, g = 0, h = 0;
  size_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;   // float offsets into the parameter arena
  int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0;
  bool has_src1 = false, has_bias = false, has_film = false, has_res = false, has_res1 = false, has_ostat = false,
       has_st1 = false, has_x2 = false, has_x21 = false;
  size_t ss_rel = 0, mr_rel = 0;   // GroupNorm tables: offset inside the scale/shift region (0 in inference)
  bool has_mr = false, has_drop = false;
  unsigned drop_key = 0;
  // the NEXT op's GroupNorm fold done by this op's last kernel (plan option fold_fuse; FoldTail in sr3_common.h): a split-K conv's
  // reduce, or the stand-alone statistics pass
  bool fold_fused = false, f_has_o = false, f_has_mr = false;
  int f_Ctot = 0, f_coff = 0, f_oC = 0, f_oT = 0, f_ooff = 0;
  size_t f_ostat = 0, f_gamma = 0, f_beta = 0, f_ss_rel = 0, f_mr_rel = 0;
  ConvParams cp;                   // OP_CONV geometry (pointers filled at launch)
  int tile_cfg = 0, ksplit = 0;
  size_t wino_off = 0;             // tile_cfg 11: float offset of this conv's transformed filters in the derived buffer
  // plan option fork_side (inference plans): side_id >= 0 -- this op depends on nothing the ops between it and its consumer write, so it
  // is launched on the plan's side stream (forked from the caller's stream by an event, joined by event side_id); wait_id >= 0 -- the
  // caller's stream waits for join event wait_id before this op (the consumer).  Inside a stream capture the pair becomes a parallel
  // branch of the graph.
  int side_id = -1, wait_id = -1;
  bool has_wsplit = false;         // im2col SPLIT tile: its weights pre-split into bf16 planes sit in the derived buffer ...
  size_t wsplit_off = 0;           // ... at this float offset (ConvParams::w_split)
};

struct Tap { std::string name; size_t off; int C, H, W; };

// where the fixed regions of a compiled forward live inside the workspace
struct Regions {
  const std::vector<Op>* ops;
  size_t stats_off, ss_off, mr_off, temb_off, film_off, scratch_off, scratch_bytes;
};

// one logical operation of the forward, recorded for the backward walk (train mode)
enum RecKind { R_CONV_IN, R_CONV, R_ATTN, R_CONV_OUT };
struct Rec {
  int kind = R_CONV;
  int x0 = -1, x1 = -1, out = -1, r0 = -1, r1 = -1, q0 = -1, q1 = -1;   // tensor handles
  int ksize = 3, stride = 1, ups = 0, act = 0, film_row = -1;
  size_t w = 0, bias = 0, qw = 0, qb = 0, gamma = 0, beta = 0;         // parameter arena offsets
  bool has_bias = false, has_q = false;
  size_t ss_off = 0, mr_off = 0;     // persistent GroupNorm tables of this conv's prologue (bytes in the workspace)
  int qkv = -1, o = -1;              // attention
  bool has_drop = false;             // block2 conv: train-mode dropout on its activated input
  unsigned drop_key = 0;
};

struct DropCfg { unsigned seed, thresh; float scale; };
inline unsigned drop_layer_seed(unsigned seed, unsigned key) { return seed + (key + 1u) * 0x632BE5ABu; }

}  // namespace sr3

using namespace sr3;

struct sr3_plan {
  sr3_unet_desc d;
  std::vector<sr3_param_info> params;
  std::map<std::string, int> pindex;
  size_t param_floats = 0;
  int F = 0;
  size_t film_w = 0, film_b = 0;
  size_t emb_w1 = 0, emb_b1 = 0, emb_w2 = 0, emb_b2 = 0;
  std::vector<Layer> downs, mid, ups;
  size_t fin_gn_w = 0, fin_gn_b = 0, fin_w = 0, fin_b = 0;
  int fin_cin = 0, out_ch = 0;
  // options
  int fuse_stats = 1, fuse_res = 1, tile_cfg = 0, ksplit = 0, keep_all = 0, split_bf16 = 0;
  int winograd = 1;          // 3x3 stride-1 convs of the inference plan on the Winograd F(2x2,3x3) kernel (conv3x3_wino.hip)
  int wino2 = 1;             // wino_split convs of the one-image tile without dropout on the two-workgroups-per-CU kernel (conv3x3_wino2.hip,
                             // 8 x 16 pixel tile; round 6).  0: the 8-wave kernel of conv3x3_wino.hip everywhere
  int wino_split = 1;        // ... on its 3 x bf16 split instantiation (bf16 MFMA, fp32-class results; gated by tests/: error not
                             // above the fp32-MFMA instantiation's on every layer shape, 2000-step drift) where that exists: the
                             // one-image tile of the inference plan.  0: the exact-fp32 MFMA instantiation everywhere
  int wino_split8 = 1;       // ... and the four-image tile of the 8x8 maps too (round 5; no dropout form: the training forward keeps fp32)
  int wgrad_split = 1;       // training: weight gradients of the layers with > 64 channels either side on the split kernel (wgrad.hip; round 5)
  int attn_split = 1;        // SelfAttention's two contractions on the 3 x bf16 split instantiation of k_attention_v2 (round 5)
  int gemm_tile = 0;         // A/B knob: force this im2col tile (1-4) on every conv of that kernel; 0 = conv_pick's choice
  int gemm_split = 1;        // the im2col kernel (1x1 and stride-2 convs) on its 3 x bf16 split instantiations (conv_igemm.hip)
  int gemm_wpre = 0;         // 1: ... reading their weights pre-split AND in MFMA fragment order from the derived buffer, straight from global
                             // memory (round 6's form: no LDS staging of the weights; tiles 18-21 at the ABI).  Measured SLOWER in the forward
                             // again (1.50-1.52 vs 1.44-1.46 ms over the 33 launches, profiles/r06_gemm_wpre_fragment_major.txt: every wave
                             // fetches its own fragments, 3x the weight traffic of one staged copy per workgroup, and the A staging that
                             // bounds these launches is unchanged) -- off by default, an A/B knob.  0: weights split while staged (14-17)
  int fold_fuse = 1;         // the GroupNorm fold of a consumer done by the kernel that completes its (last) source where that is a split-K
                             // reduce or a stand-alone statistics pass (k_rows_fold; round 6): 32 of the 61 fold launches of the C2 forward
  int gemm2 = 1;             // 1x1 stride-1 convs (res_conv, the attention projections) on the plain GEMM kernel of gemm1x1.hip where it fits
                             // (Cout % 128 == 0, channels % 32 == 0, rows % 64 == 0): pre-split weights in fragment order read straight from
                             // global memory, A rows split once per 128 output channels, staging arithmetic hand-placed between the MFMAs
                             // (round 6; 27 of the 33 launches of the C2 forward: 1.09 -> 0.80 ms, profiles/r06_gemm1x1.txt).  0: the im2col kernel
  int gemm_s2 = 1;           // ... and Downsample's 3x3 stride-2 convs on that kernel's stride-2 form (needs gemm2; Cout % 128 == 0; last session of
                             // round 6: the three launches of the C2 forward 171 -> 118 us)
  int gemm_n64 = 1;          // ... and the layers with Cout % 128 != 0 (Cout % 64 == 0: the res_convs of the 128 x 128 level, Downsample 64 -> 64) on
                             // its 64-column tile (waves 2 x 2); 0: they keep the im2col kernel
  int fork_side = 0;         // res_conv (and the embedding MLP) on a side stream beside block1's conv: see Op::side_id; A/B knob
  hipStream_t side_stream = nullptr;          // fork_side: created at the first forked forward, on the device current then
  std::vector<hipEvent_t> fork_ev, join_ev;   // one pair per forked op of the compiled forward
  ~sr3_plan() {
    for (hipEvent_t e : fork_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : join_ev) (void)hipEventDestroy(e);
    if (side_stream) (void)hipStreamDestroy(side_stream);
  }
  // derived weights: U = G g G^T of every 3x3 stride-1 conv, fragment-major (caller-owned buffer, bound by pointer)
  struct Derived { size_t w; int Cout, Cin; size_t off; };
  std::vector<Derived> derived;
  std::map<size_t, size_t> derived_of;     // weight arena offset -> float offset in the derived buffer
  // ... and, plan option gemm_split, the 1x1 / stride-2 weights of the im2col SPLIT tiles as three bf16 planes (conv_igemm.hip)
  struct WSplit { size_t w; int Cout, taps, Cin; size_t off; };
  std::vector<WSplit> wsplits;
  std::map<size_t, size_t> wsplit_of;      // weight arena offset -> float offset in the derived buffer
  size_t derived_floats = 0;
  float* derived_ptr = nullptr;
  size_t derived_bound_bytes = 0;
  const float* derived_from = nullptr;   // the arena sr3_plan_prepare_derived last ran on (null: never / invalidated)
  int loss_l2 = 0;           // training loss: 0 = L1 (sum), 1 = L2 (sum)  (set_loss, diffusion.py:84-90)
  // compiled forward
  int built_batch = -1;
  int built_cond = -1;
  std::vector<Op> ops;
  std::vector<Tap> taps;
  size_t ws_bytes = 0;
  size_t stats_off = 0, stats_bytes = 0, ss_off = 0, temb_off = 0, film_off = 0, scratch_off = 0, scratch_bytes = 0;
  double flops = 0;
  // ---- training step (train_plan.hip) ----
  int train_batch = -1, train_cond = -1;
  std::vector<Op> tops;              // forward ops in train mode (no buffer reuse, persistent GN tables)
  std::vector<sr3::Tensor> ttens;    // tensor table of the train forward
  std::vector<sr3::Rec> recs;
  size_t t_act_bytes = 0;            // activations; gradients mirror them at +t_act_bytes
  size_t t_stats_off = 0, t_gn_off = 0, t_temb_off = 0, t_film_off = 0, t_scratch_off = 0, t_scratch_bytes = 0;
  size_t t_dA_off = 0, t_z_off = 0, t_dq_off = 0, t_wt_off = 0, t_slab_off = 0, t_part_off = 0, t_gs_off = 0;
  size_t t_dfilm_off = 0, t_misc_off = 0, t_xnoisy_off = 0, t_eps_off = 0, t_geps_off = 0, t_inpad_off = 0, t_dwtmp_off = 0;
  size_t t_ws_bytes = 0, t_embscr_off = 0, t_a_off = 0;
  size_t t_wu_off = 0, t_wu_bytes = 0;   // Winograd-transformed filters of the data-gradient conv being run
  std::vector<size_t> t_unproc_max;  // [ri]: max parameter end offset still unwritten before record ri-1 is processed
  int t_final_x = -1;                // tensor handle feeding the output Block
  size_t t_final_ss = 0, t_final_mr = 0;
  int t_conv_in_out = -1;
};


namespace sr3 {
struct Builder;
Regions infer_regions(const sr3_plan* P);
int run_forward(sr3_plan* P, const Regions& R, const float* x, const float* cond, int cond_channels, const float* level,
                const int64_t* tstep, const float* freq, const float* level_table, const int* step_dev,
                const float* params, char* ws, float* eps_out, int B, hipStream_t st, hipEvent_t* ev, hipEvent_t* mid,
                const DropCfg* drop = nullptr, const StepFuse* fuse = nullptr);
int build_train(sr3_plan* P, int B, int cond_channels);
void layout_derived(sr3_plan* P);
}  // namespace sr3
