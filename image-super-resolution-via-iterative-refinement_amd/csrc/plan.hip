// UNet plan: topology, parameter arena, activation plan and the forward driver.
//
// Mirrors UNet.__init__ / UNet.forward of the reference (model/sr3_modules/unet.py:162-259,
// model/ddpm_modules/unet.py:148-243) as a static op list over the HIP kernels of this library.
// Nothing here is a translation of the reference modules: the forward is compiled once per batch
// size into a flat list of kernel launches over a liveness-planned NHWC workspace.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "plan_internal.h"
#include "train.h"

namespace sr3 {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  (void)hipGetLastError();
  return (int)e > 0 ? (int)e : 1;
}
const char* last_error() { return g_err; }

}  // namespace sr3

namespace sr3 {

// ---------------------------------------------------------------------------------------------
// parameter table
// ---------------------------------------------------------------------------------------------
static size_t add_param(sr3_plan* P, const std::string& name, int ndim, const int* shape, int pack, size_t* cursor) {
  sr3_param_info pi;
  memset(&pi, 0, sizeof(pi));
  snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
  pi.ndim = ndim;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { pi.shape[i] = shape[i]; n *= (size_t)shape[i]; }
  pi.pack = pack;
  pi.numel = n;
  *cursor = (*cursor + 3) & ~(size_t)3;   // 16-byte alignment of every tensor
  pi.offset = *cursor;
  *cursor += n;
  P->pindex[name] = (int)P->params.size();
  P->params.push_back(pi);
  return pi.offset;
}
static size_t add_vec(sr3_plan* P, const std::string& name, int n, size_t* cur) {
  int s[1] = {n};
  return add_param(P, name, 1, s, 0, cur);
}
static size_t add_mat(sr3_plan* P, const std::string& name, int o, int i, size_t* cur) {
  int s[2] = {o, i};
  return add_param(P, name, 2, s, 0, cur);
}
static size_t add_conv(sr3_plan* P, const std::string& name, int o, int i, int k, size_t* cur) {
  int s[4] = {o, i, k, k};
  return add_param(P, name, 4, s, k == 1 ? 0 : 1, cur);
}

static bool in_list(const int* v, int n, int x) {
  for (int i = 0; i < n; ++i) if (v[i] == x) return true;
  return false;
}

static int build_structure(sr3_plan* P) {
  const sr3_unet_desc& d = P->d;
  const int inner = d.inner_channel;
  const bool ddpm = d.variant == SR3_VARIANT_DDPM;
  if (inner <= 0 || (inner & 3)) { set_error("inner_channel must be a positive multiple of 4 (got %d)", inner); return SR3_E_UNSUPPORTED; }
  if (d.n_mults < 1 || d.n_mults > 8 || d.n_attn_res < 0 || d.n_attn_res > 8) { set_error("bad n_mults/n_attn_res"); return SR3_E_BADARG; }
  if (d.norm_groups <= 0) { set_error("norm_groups must be > 0"); return SR3_E_BADARG; }
  if (d.in_channel <= 0 || d.in_channel > 16) { set_error("in_channel %d unsupported", d.in_channel); return SR3_E_UNSUPPORTED; }
  const int out_ch = d.out_channel > 0 ? d.out_channel : d.in_channel;
  if (out_ch > 4) { set_error("out_channel %d > 4 unsupported", out_ch); return SR3_E_UNSUPPORTED; }
  if ((d.image_size >> (d.n_mults - 1)) < 1 || (d.image_size & ((1 << (d.n_mults - 1)) - 1))) {
    set_error("image_size %d not divisible by 2^%d", d.image_size, d.n_mults - 1);
    return SR3_E_BADARG;
  }
  P->out_ch = out_ch;

  // ---- pass 1: topology (same walk as UNet.__init__) ----
  struct Proto { int kind; int cin, cout, skip; bool attn; };
  std::vector<Proto> pd, pm, pu;
  std::vector<int> feat;
  int pre = inner, now_res = d.image_size;
  feat.push_back(pre);
  pd.push_back({0, d.in_channel, inner, 0, false});
  for (int ind = 0; ind < d.n_mults; ++ind) {
    const bool is_last = ind == d.n_mults - 1;
    const bool use_attn = in_list(d.attn_res, d.n_attn_res, now_res);
    const int cm = inner * d.channel_mults[ind];
    for (int r = 0; r < d.res_blocks; ++r) {
      pd.push_back({1, pre, cm, 0, use_attn});
      feat.push_back(cm);
      pre = cm;
    }
    if (!is_last) {
      pd.push_back({2, pre, pre, 0, false});
      feat.push_back(pre);
      now_res /= 2;
    }
  }
  pm.push_back({1, pre, pre, 0, true});
  pm.push_back({1, pre, pre, 0, false});
  for (int ind = d.n_mults - 1; ind >= 0; --ind) {
    const bool is_last = ind < 1;
    const bool use_attn = in_list(d.attn_res, d.n_attn_res, now_res);
    const int cm = inner * d.channel_mults[ind];
    for (int r = 0; r < d.res_blocks + 1; ++r) {
      const int skip = feat.back();
      feat.pop_back();
      pu.push_back({1, pre + skip, cm, skip, use_attn});
      pre = cm;
    }
    if (!is_last) {
      pu.push_back({3, pre, pre, 0, false});
      now_res *= 2;
    }
  }
  P->fin_cin = pre;

  // ---- pass 2: arena.  FiLM projections first (contiguous => one GEMV for all blocks) ----
  size_t cur = 0;
  int F = 0;
  auto count_f = [&](const std::vector<Proto>& v) { for (auto& p : v) if (p.kind == 1) F += p.cout; };
  count_f(pd); count_f(pm); count_f(pu);
  P->F = F;
  P->film_w = cur; cur += (size_t)F * inner;
  cur = (cur + 3) & ~(size_t)3;
  P->film_b = cur; cur += (size_t)F;
  int frow = 0;

  const char* emb = ddpm ? "time_mlp" : "noise_level_mlp";
  P->emb_w1 = add_mat(P, std::string(emb) + ".1.weight", 4 * inner, inner, &cur);
  P->emb_b1 = add_vec(P, std::string(emb) + ".1.bias", 4 * inner, &cur);
  P->emb_w2 = add_mat(P, std::string(emb) + ".3.weight", inner, 4 * inner, &cur);
  P->emb_b2 = add_vec(P, std::string(emb) + ".3.bias", inner, &cur);

  auto emit = [&](const std::vector<Proto>& protos, const char* grp, std::vector<Layer>& out) {
    for (size_t i = 0; i < protos.size(); ++i) {
      const Proto& pr = protos[i];
      Layer L;
      L.kind = pr.kind;
      L.cin = pr.cin;
      L.cout = pr.cout;
      char nm[64];
      snprintf(nm, sizeof(nm), "%s.%zu", grp, i);
      L.name = nm;
      L.w = L.b = 0;
      std::string n = nm;
      if (pr.kind == 0) {
        L.w = add_conv(P, n + ".weight", pr.cout, pr.cin, 3, &cur);
        L.b = add_vec(P, n + ".bias", pr.cout, &cur);
      } else if (pr.kind == 2 || pr.kind == 3) {
        L.w = add_conv(P, n + ".conv.weight", pr.cout, pr.cin, 3, &cur);
        L.b = add_vec(P, n + ".conv.bias", pr.cout, &cur);
      } else {
        ResLayer& R = L.res;
        R.name = n;
        R.cin = pr.cin; R.cout = pr.cout; R.skip = pr.skip; R.attn = pr.attn;
        R.film_off = frow;
        std::string rb = n + ".res_block";
        // FiLM rows live in the contiguous block; the table entries point into it
        {
          sr3_param_info pi;
          memset(&pi, 0, sizeof(pi));
          std::string wn = rb + (ddpm ? ".mlp.1.weight" : ".noise_func.noise_func.0.weight");
          std::string bn = rb + (ddpm ? ".mlp.1.bias" : ".noise_func.noise_func.0.bias");
          snprintf(pi.name, sizeof(pi.name), "%s", wn.c_str());
          pi.ndim = 2; pi.shape[0] = pr.cout; pi.shape[1] = inner; pi.pack = 0;
          pi.numel = (size_t)pr.cout * inner; pi.offset = P->film_w + (size_t)frow * inner;
          P->pindex[wn] = (int)P->params.size(); P->params.push_back(pi);
          memset(&pi, 0, sizeof(pi));
          snprintf(pi.name, sizeof(pi.name), "%s", bn.c_str());
          pi.ndim = 1; pi.shape[0] = pr.cout; pi.pack = 0; pi.numel = (size_t)pr.cout; pi.offset = P->film_b + frow;
          P->pindex[bn] = (int)P->params.size(); P->params.push_back(pi);
        }
        frow += pr.cout;
        R.gn1_w = add_vec(P, rb + ".block1.block.0.weight", pr.cin, &cur);
        R.gn1_b = add_vec(P, rb + ".block1.block.0.bias", pr.cin, &cur);
        R.c1_w = add_conv(P, rb + ".block1.block.3.weight", pr.cout, pr.cin, 3, &cur);
        R.c1_b = add_vec(P, rb + ".block1.block.3.bias", pr.cout, &cur);
        R.gn2_w = add_vec(P, rb + ".block2.block.0.weight", pr.cout, &cur);
        R.gn2_b = add_vec(P, rb + ".block2.block.0.bias", pr.cout, &cur);
        R.c2_w = add_conv(P, rb + ".block2.block.3.weight", pr.cout, pr.cout, 3, &cur);
        R.c2_b = add_vec(P, rb + ".block2.block.3.bias", pr.cout, &cur);
        R.has_rc = pr.cin != pr.cout;
        R.rc_w = R.rc_b = 0;
        if (R.has_rc) {
          R.rc_w = add_conv(P, rb + ".res_conv.weight", pr.cout, pr.cin, 1, &cur);
          R.rc_b = add_vec(P, rb + ".res_conv.bias", pr.cout, &cur);
        }
        R.an_w = R.an_b = R.qkv_w = R.ao_w = R.ao_b = 0;
        if (pr.attn) {
          std::string at = n + ".attn";
          R.an_w = add_vec(P, at + ".norm.weight", pr.cout, &cur);
          R.an_b = add_vec(P, at + ".norm.bias", pr.cout, &cur);
          R.qkv_w = add_conv(P, at + ".qkv.weight", 3 * pr.cout, pr.cout, 1, &cur);
          R.ao_w = add_conv(P, at + ".out.weight", pr.cout, pr.cout, 1, &cur);
          R.ao_b = add_vec(P, at + ".out.bias", pr.cout, &cur);
        }
      }
      out.push_back(L);
    }
  };
  emit(pd, "downs", P->downs);
  emit(pm, "mid", P->mid);
  emit(pu, "ups", P->ups);
  P->fin_gn_w = add_vec(P, "final_conv.block.0.weight", P->fin_cin, &cur);
  P->fin_gn_b = add_vec(P, "final_conv.block.0.bias", P->fin_cin, &cur);
  P->fin_w = add_conv(P, "final_conv.block.3.weight", out_ch, P->fin_cin, 3, &cur);
  P->fin_b = add_vec(P, "final_conv.block.3.bias", out_ch, &cur);
  P->param_floats = (cur + 3) & ~(size_t)3;
  layout_derived(P);

  // every GroupNorm must divide
  auto chk = [&](int c) { return c % d.norm_groups == 0; };
  bool ok = chk(P->fin_cin);
  for (auto* v : {&P->downs, &P->mid, &P->ups})
    for (auto& L : *v)
      if (L.kind == 1) ok = ok && chk(L.res.cin) && chk(L.res.cout);
  if (!ok) { set_error("a GroupNorm channel count is not divisible by norm_groups=%d", d.norm_groups); return SR3_E_BADARG; }
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// activation plan (first-fit free list, byte offsets; sizes scale with batch)
// ---------------------------------------------------------------------------------------------
struct Arena {
  struct Blk { size_t off, size; };
  std::vector<Blk> free_list;
  size_t top = 0, high = 0;
  size_t alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (size_t i = 0; i < free_list.size(); ++i) {
      if (free_list[i].size >= bytes) {
        const size_t off = free_list[i].off;
        free_list[i].off += bytes;
        free_list[i].size -= bytes;
        if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
        return off;
      }
    }
    const size_t off = top;
    top += bytes;
    if (top > high) high = top;
    return off;
  }
  void release(size_t off, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    free_list.push_back({off, bytes});
    std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < free_list.size();) {
      if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
        free_list[i].size += free_list[i + 1].size;
        free_list.erase(free_list.begin() + i + 1);
      } else {
        ++i;
      }
    }
    if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
      top = free_list.back().off;
      free_list.pop_back();
    }
  }
};

struct Builder {
  sr3_plan* P;
  int B;
  Arena act;
  std::vector<Tensor> T;          // tensor table; handles are indices (shared state, no copies)
  size_t stats_cursor = 0;
  size_t max_scratch = 0;
  int max_cin = 0;
  double flops = 0;
  std::vector<Op>& ops;
  bool train = false;
  size_t gn_cursor = 0, mr_cursor = 0;     // train: persistent per-GroupNorm tables
  size_t cur_ss = 0, cur_mr = 0;           // tables written by the most recent fold
  size_t cur_gamma = 0, cur_beta = 0;
  size_t max_dA = 0, max_wt = 0, max_slab = 0, max_part = 0, max_z = 0, max_dq = 0;
  Builder(sr3_plan* p, int b, bool tr = false) : P(p), B(b), ops(tr ? p->tops : p->ops), train(tr) {}

  int make(int C, int H, int W) {
    Tensor t;
    t.C = C; t.H = H; t.W = W;
    t.bytes = (size_t)B * H * W * C * sizeof(float);
    t.off = act.alloc(t.bytes);
    t.valid = true;
    T.push_back(t);
    return (int)T.size() - 1;
  }
  void drop(int h) {
    if (h < 0 || !T[h].valid) return;
    if (train) return;                     // the backward needs every activation
    if (!P->keep_all) act.release(T[h].off, T[h].bytes);
    T[h].valid = false;
  }
  void stat_slot(int h, int parts) {
    Tensor& t = T[h];
    t.stat_T = parts;
    t.stat_off = stats_cursor;
    stats_cursor += (size_t)B * parts * t.C * 2 * sizeof(double);
    t.stats_done = true;
  }
  void ensure_stats(int h) {
    Tensor& t = T[h];
    if (t.stats_done) return;
    stat_slot(h, chan_stats_slices(B, t.H * t.W, t.C));
    Op o; o.kind = OP_STATS;
    o.a = t.off; o.b = t.stat_off; o.i0 = t.H * t.W; o.i1 = t.C;
    ops.push_back(o);
    last_stats_op = (int)ops.size() - 1; last_stats_h = h;
  }
  int last_conv_op = -1, last_conv_out = -1, last_stats_op = -1, last_stats_h = -1;
  // plan option fold_fuse: can the kernel that completes `fresh` (one of x0 / x1) also do this fold?  Fills the op's FoldTail fields.
  bool fuse_fold_into(Op& L, int fresh, int x0, int x1, size_t gamma, size_t beta, size_t ss_rel, size_t mr_rel, bool has_mr) {
    const int other = fresh == x0 ? x1 : x0;
    const int Cf = T[x0].C + (x1 >= 0 ? T[x1].C : 0);
    const int c_off = fresh == x0 ? 0 : T[x0].C;
    if (!fold_tail_fits(T[fresh].C, Cf, c_off, P->d.norm_groups)) return false;
    if (other >= 0 && !T[other].stats_done) return false;
    L.fold_fused = true;
    L.f_Ctot = Cf; L.f_coff = c_off;
    L.f_has_o = other >= 0;
    if (other >= 0) { L.f_ostat = T[other].stat_off; L.f_oC = T[other].C; L.f_oT = T[other].stat_T; L.f_ooff = fresh == x0 ? T[x0].C : 0; }
    L.f_gamma = gamma; L.f_beta = beta; L.f_ss_rel = ss_rel; L.f_mr_rel = mr_rel; L.f_has_mr = has_mr;
    T[fresh].stat_T = 1;              // the fused kernel holds whole-image sums: one partial per image
    return true;
  }
  void fold(int x0, int x1, size_t gamma, size_t beta) {
    const int Cfu = T[x0].C + (x1 >= 0 ? T[x1].C : 0);
    size_t ss_rel = 0, mr_rel = 0;
    auto take_slots = [&]() {
      if (train) {
        ss_rel = gn_cursor; gn_cursor += ((size_t)B * Cfu * 2 * sizeof(float) + 255) & ~(size_t)255;
        mr_rel = mr_cursor; mr_cursor += ((size_t)B * P->d.norm_groups * 2 * sizeof(float) + 255) & ~(size_t)255;
      }
      cur_ss = ss_rel; cur_mr = mr_rel; cur_gamma = gamma; cur_beta = beta;
      max_cin = std::max(max_cin, Cfu);
    };
    if (P->fold_fuse && !ops.empty()) {
      // (a) the op just emitted is a split-K conv whose reduce writes the statistics of x0 / x1: its reduce folds too
      Op& L = ops.back();
      if (last_conv_op == (int)ops.size() - 1 && L.kind == OP_CONV && L.ksplit > 1 && L.has_ostat && (last_conv_out == x0 || last_conv_out == x1)) {
        take_slots();
        if (fuse_fold_into(L, last_conv_out, x0, x1, gamma, beta, ss_rel, mr_rel, train)) return;
        if (train) { gn_cursor = ss_rel; mr_cursor = mr_rel; }       // (not fused: the slots are taken again below)
      }
    }
    ensure_stats(x0);
    if (x1 >= 0) ensure_stats(x1);
    if (P->fold_fuse && !ops.empty() && last_stats_op == (int)ops.size() - 1 && (last_stats_h == x0 || last_stats_h == x1)) {
      // (b) ... or a stand-alone statistics pass of x0 / x1: it folds too
      take_slots();
      if (fuse_fold_into(ops.back(), last_stats_h, x0, x1, gamma, beta, ss_rel, mr_rel, train)) return;
      if (train) { gn_cursor = ss_rel; mr_cursor = mr_rel; }
    }
    Op o; o.kind = OP_FOLD;
    o.a = T[x0].stat_off; o.i0 = T[x0].C; o.i3 = T[x0].stat_T;
    o.has_st1 = x1 >= 0;
    if (x1 >= 0) { o.b = T[x1].stat_off; o.i1 = T[x1].C; o.i4 = T[x1].stat_T; }
    o.i2 = T[x0].H * T[x0].W;
    o.p0 = gamma; o.p1 = beta;
    const int Cf = T[x0].C + (x1 >= 0 ? T[x1].C : 0);
    if (train) {
      o.ss_rel = gn_cursor; gn_cursor += ((size_t)B * Cf * 2 * sizeof(float) + 255) & ~(size_t)255;
      o.mr_rel = mr_cursor; mr_cursor += ((size_t)B * P->d.norm_groups * 2 * sizeof(float) + 255) & ~(size_t)255;
      o.has_mr = true;
    }
    cur_ss = o.ss_rel; cur_mr = o.mr_rel; cur_gamma = gamma; cur_beta = beta;
    ops.push_back(o);
    max_cin = std::max(max_cin, Cf);
  }
  // every 3x3 stride-1 conv the Winograd kernel covers runs on it (plan option `winograd`, default on), train-mode dropout
  // convs included; an explicit tile_cfg / split_bf16 keep the direct halo kernels (the fused res_conv segment has no
  // Winograd form: res_conv is then its own 1x1 GEMM)
  bool wino_ok(const ConvParams& c, size_t w, bool has_x2, bool has_drop) {
    if (!P->winograd || P->tile_cfg != 0 || P->split_bf16 || has_x2) return false;
    if (has_drop && (c.C1 != 0 || c.ups != 0 || c.act == 0)) return false;    // the dropout form: single source, no upsampling
    if (c.ksize != 3 || c.stride != 1 || !P->derived_of.count(w)) return false;
    WinoGeom wg;
    return wino_geometry(c, &wg);
  }
  // generic conv over the virtual concat (x0|x1); residual is the concat view (r0|r1)
  int conv(int x0, int x1, int Cout, int ksize, int stride, int ups, int act_mode, size_t w, bool has_bias,
           size_t bias, int film_row, int r0, int r1, bool want_stats, int q0 = -1, int q1 = -1, size_t qw = 0,
           size_t qb = 0, int drop_key = -1) {
    const int C0 = T[x0].C, C1 = x1 >= 0 ? T[x1].C : 0;
    const int Hi = T[x0].H << ups, Wi = T[x0].W << ups;
    const int pad = ksize / 2;
    const int Ho = (Hi + 2 * pad - ksize) / stride + 1, Wo = (Wi + 2 * pad - ksize) / stride + 1;
    const int out = make(Cout, Ho, Wo);
    Op o; o.kind = OP_CONV;
    ConvParams& c = o.cp;
    memset(&c, 0, sizeof(c));
    c.C0 = C0; c.C1 = C1;
    c.B = B; c.Hs = T[x0].H; c.Ws = T[x0].W; c.ups = ups; c.stride = stride; c.ksize = ksize;
    c.Ho = Ho; c.Wo = Wo; c.Cout = Cout; c.act = act_mode;
    c.film_stride = P->F;
    c.RC0 = r0 >= 0 ? T[r0].C : 0; c.RC1 = r1 >= 0 ? T[r1].C : 0;
    c.ksplit = 1;
    o.a = T[x0].off; o.has_src1 = x1 >= 0; if (x1 >= 0) o.b = T[x1].off;
    o.p0 = w; o.has_bias = has_bias; o.p1 = bias;
    o.has_film = film_row >= 0; o.i0 = film_row;
    o.has_res = r0 >= 0; if (r0 >= 0) o.c = T[r0].off;
    o.has_res1 = r1 >= 0; if (r1 >= 0) o.d = T[r1].off;
    o.e = T[out].off;
    o.tile_cfg = P->tile_cfg; o.ksplit = P->ksplit;
    o.ss_rel = act_mode ? cur_ss : 0;
    o.has_drop = train && drop_key >= 0; o.drop_key = (unsigned)(drop_key >= 0 ? drop_key : 0);
    // opt-in: the 3 x bf16 split MFMA instantiation of the 8-wave tile wherever it fits (inference plans only)
    {
      HaloGeom sg;
      if (P->split_bf16 && !train && o.tile_cfg == 0 && ksize == 3 && stride == 1 && Cout > 64 && halo_geometry(c, 10, &sg))
        o.tile_cfg = 10;
    }
    if (wino_ok(c, w, q0 >= 0, o.has_drop)) {
      o.tile_cfg = 11;
      o.wino_off = P->derived_of[w];
      // plan option wino_split: the 3 x bf16 split instantiation where it exists (the one-image tile; training plan and
      // train-mode dropout included); its filters sit behind the conv's fp32 ones
      WinoGeom wg;
      if (P->wino_split && wino_geometry(c, &wg) && (wg.NB == 1 || (P->wino_split8 && !o.has_drop))) {
        c.wino_split = (P->wino2 && !o.has_drop && wg.NB == 1) ? 2 : 1;        // (2: the 8 x 16 tile of conv3x3_wino2.hip)
        o.wino_off += wino_weight_floats(Cout, C0 + C1);
      }
    }
    // read by conv_pick / conv_forward only when the conv lands on the im2col kernel; the 9-tap layers with Cout <= 64 (Downsample
    // of the first level) stay on the fp32 MFMA, which is faster there (67 vs 81 us in the forward)
    c.igemm_split = (P->gemm_split && !(ksize == 3 && Cout <= 64)) ? 1 : 0;
    // (... unless the plain GEMM kernel takes the layer on its 64-column tile: plan options gemm2 + gemm_s2 + gemm_n64, below)
    if (P->gemm_split && P->gemm2 && P->gemm_s2 && P->gemm_n64 && ksize == 3 && stride == 2 && P->wsplit_of.count(w) && gemm1x1_fits(c, 2)) c.igemm_split = 1;
    conv_pick(c, o.tile_cfg, o.ksplit);
    if (P->gemm_tile >= 1 && P->gemm_tile <= 4 && o.tile_cfg >= 1 && o.tile_cfg <= 4 && P->tile_cfg == 0) {
      o.tile_cfg = P->gemm_tile; o.ksplit = P->ksplit;       // A/B knob: one im2col tile for every conv of that kernel
      conv_pick(c, o.tile_cfg, o.ksplit);
    }
    if (c.igemm_split && o.tile_cfg >= 1 && o.tile_cfg <= 4 && P->gemm2 && P->tile_cfg == 0 && P->gemm_tile == 0 && P->wsplit_of.count(w)) {
      // plan option gemm2: 1x1 stride-1 convs -- and Downsample's 3x3 stride-2 ones -- the plain GEMM kernel fits (gemm1x1.hip)
      if (gemm1x1_fits(c, 2) && (ksize == 1 || P->gemm_s2) && (!(Cout & 127) || P->gemm_n64)) {
        o.tile_cfg = 22; o.ksplit = P->ksplit;
        conv_pick(c, o.tile_cfg, o.ksplit);
        o.has_wsplit = true; o.wsplit_off = P->wsplit_of[w];
      }
    }
    if (c.igemm_split && o.tile_cfg >= 1 && o.tile_cfg <= 4 && P->gemm_wpre && P->wsplit_of.count(w)) {
      o.has_wsplit = true; o.wsplit_off = P->wsplit_of[w];       // the SPLIT tile reads its weights pre-split from the derived buffer
    }
    if (o.has_drop && o.tile_cfg == 9) {       // no dropout instantiation of the 8-wave tile
      o.tile_cfg = 5; o.ksplit = P->ksplit;
      conv_pick(c, o.tile_cfg, o.ksplit);
    }
    // dropout convs on the 256x64 tile run its 8-wave form (conv3x3_halo_forward); SR3_DROP_CFG5 forces the
    // half-empty 128x128 tile instead (A/B knob)
    static const bool use5 = getenv("SR3_DROP_CFG5") != nullptr;
    if (o.has_drop && o.tile_cfg == 6 && use5) {
      o.tile_cfg = 5; o.ksplit = P->ksplit;
      conv_pick(c, o.tile_cfg, o.ksplit);
    }
    // opt-in: run the halo-tile convs on the 3 x bf16 split MFMA instantiations (inference plans only)
    if (P->split_bf16 && !train && o.tile_cfg == 5) o.tile_cfg = 7;
    else if (P->split_bf16 && !train && o.tile_cfg == 9) o.tile_cfg = 10;
    else if (P->split_bf16 && !train && o.tile_cfg == 6) o.tile_cfg = 8;
    if (train) {
      Rec r;
      r.kind = R_CONV; r.x0 = x0; r.x1 = x1; r.out = out; r.r0 = r0; r.r1 = r1; r.q0 = q0; r.q1 = q1;
      r.ksize = ksize; r.stride = stride; r.ups = ups; r.act = act_mode; r.film_row = film_row;
      r.w = w; r.bias = bias; r.has_bias = has_bias; r.qw = qw; r.qb = qb; r.has_q = q0 >= 0;
      r.gamma = cur_gamma; r.beta = cur_beta; r.ss_off = cur_ss; r.mr_off = cur_mr;
      r.has_drop = o.has_drop; r.drop_key = o.drop_key;
      P->recs.push_back(r);
      // scratch the backward of this conv needs
      const size_t cin = (size_t)(C0 + C1);
      max_dA = std::max(max_dA, (size_t)B * Hi * Wi * cin * sizeof(float));
      max_wt = std::max(max_wt, (size_t)Cout * ksize * ksize * cin * sizeof(float));
      if (q0 >= 0) {
        const size_t cq = (size_t)T[q0].C + (q1 >= 0 ? T[q1].C : 0);
        max_dq = std::max(max_dq, (size_t)B * Ho * Wo * cq * sizeof(float));
        max_wt = std::max(max_wt, (size_t)Cout * cq * sizeof(float));
      }
      if (stride == 2) max_z = std::max(max_z, (size_t)B * 4 * Ho * Wo * Cout * sizeof(float));
    }
    if (q0 >= 0) {   // fused 1x1 segment (res_conv); caller checked can_fuse_x2()
      c.x2_C0 = T[q0].C; c.x2_C1 = q1 >= 0 ? T[q1].C : 0;
      o.has_x2 = true; o.g = T[q0].off; o.has_x21 = q1 >= 0; if (q1 >= 0) o.h = T[q1].off;
      o.p2 = qw; o.p3 = qb;
      flops += 2.0 * B * Ho * Wo * (double)Cout * (double)(c.x2_C0 + c.x2_C1);
    }
    if (o.ksplit > 1) max_scratch = std::max(max_scratch, (size_t)o.ksplit * B * Ho * Wo * Cout * sizeof(float));
    // split-K convs leave the statistics to the (cheap, small-tensor) stand-alone pass
    if (want_stats && P->fuse_stats && o.ksplit == 1 && o.tile_cfg == 11) {
      WinoGeom wg;
      wino_geometry(c, &wg);
      stat_slot(out, wino_stats_slices(wg));
      o.has_ostat = true; o.f = T[out].stat_off;
    } else if (want_stats && P->fuse_stats && o.ksplit == 1 && o.tile_cfg >= 5) {
      HaloGeom hg;
      if (halo_geometry(c, o.tile_cfg, &hg)) {
        stat_slot(out, halo_stats_slices(hg));
        o.has_ostat = true; o.f = T[out].stat_off;
      }
    } else if (want_stats && P->fuse_stats && o.ksplit > 1) {
      const int rpb = splitk_rows_per_block(c, true);      // statistics come out of the split-K reduce
      if (rpb > 0) {
        stat_slot(out, (Ho * Wo) / rpb);
        o.has_ostat = true; o.f = T[out].stat_off;
      }
    }
    ops.push_back(o);
    last_conv_op = (int)ops.size() - 1; last_conv_out = out;
    flops += 2.0 * B * Ho * Wo * (double)Cout * (double)(C0 + C1) * ksize * ksize;
    return out;
  }
  // fork_side: is res_conv its own launch in this block (block2's conv has no fused 1x1 segment)?  Same rule as can_fuse_x2, asked before h1 exists
  bool res_conv_is_own_launch(int x0, const ResLayer& R) {
    if (!P->fuse_res) return true;
    ConvParams c;
    memset(&c, 0, sizeof(c));
    c.C0 = R.cout; c.B = B; c.Hs = T[x0].H; c.Ws = T[x0].W; c.stride = 1; c.ksize = 3;
    c.Ho = T[x0].H; c.Wo = T[x0].W; c.Cout = R.cout;
    c.act = 2;
    if (wino_ok(c, R.c2_w, false, train)) return true;
    int cfg = P->tile_cfg, ks = P->ksplit;
    conv_pick(c, cfg, ks);
    return !(cfg >= 5 && ks == 1);
  }
  // can block2's conv run on the halo kernel (which can take res_conv as a second K-segment)?
  bool can_fuse_x2(int h1, int Cout, size_t w) {
    if (!P->fuse_res) return false;
    ConvParams c;
    memset(&c, 0, sizeof(c));
    c.C0 = T[h1].C; c.B = B; c.Hs = T[h1].H; c.Ws = T[h1].W; c.stride = 1; c.ksize = 3;
    c.Ho = T[h1].H; c.Wo = T[h1].W; c.Cout = Cout;
    // the Winograd kernel has no second K-segment: res_conv runs as its own 1x1 GEMM and joins as a residual (round 3: also
    // in the training plan, whose block2 conv runs the Winograd kernel's dropout instantiation)
    c.act = 2;
    if (wino_ok(c, w, false, train)) return false;
    int cfg = P->tile_cfg, ks = P->ksplit;
    conv_pick(c, cfg, ks);
    // under split-K the fused segment runs in the last split only (50 k-steps there vs 18 in the others for a
    // 1024-channel res_conv at 8x8: measured 43 TF): small-M layers keep res_conv as its own 1x1 GEMM
    return cfg >= 5 && ks == 1;
  }
  int n_side = 0;          // plan option fork_side: ops handed to the side stream so far (Op::side_id)
  int res_block(int x0, int x1, const ResLayer& R) {
    fold(x0, x1, R.gn1_w, R.gn1_b);
    // plan option fork_side (inference): res_conv reads only the block input, so it is emitted HERE -- behind the fold, in front of block1's
    // conv -- and launched on the side stream; block2's conv, which adds it as its residual, waits for it.  Only unsplit: a split-K res_conv
    // would share the slab region with block1's conv.  (The fold stays where fold_fuse looks for it: behind the op that completes x.)
    int r_side = -1, r_id = -1;
    if (R.has_rc && P->fork_side && !train && res_conv_is_own_launch(x0, R)) {
      r_side = conv(x0, x1, R.cout, 1, 1, 0, 0, R.rc_w, true, R.rc_b, -1, -1, -1, false);
      if (ops.back().kind == OP_CONV && ops.back().ksplit == 1) { r_id = n_side++; ops.back().side_id = r_id; }
    }
    const int h1 = conv(x0, x1, R.cout, 3, 1, 0, 2, R.c1_w, true, R.c1_b, R.film_off, -1, -1, true);
    fold(h1, -1, R.gn2_w, R.gn2_b);
    int out;
    if (r_side >= 0) {
      out = conv(h1, -1, R.cout, 3, 1, 0, 2, R.c2_w, true, R.c2_b, -1, r_side, -1, true, -1, -1, 0, 0, R.film_off);
      if (r_id >= 0) ops[last_conv_op].wait_id = r_id;
      drop(r_side);
    } else
    if (R.has_rc && can_fuse_x2(h1, R.cout, R.c2_w)) {
      out = conv(h1, -1, R.cout, 3, 1, 0, 2, R.c2_w, true, R.c2_b, -1, -1, -1, true, x0, x1, R.rc_w, R.rc_b, R.film_off);
    } else if (R.has_rc) {
      const int r = conv(x0, x1, R.cout, 1, 1, 0, 0, R.rc_w, true, R.rc_b, -1, -1, -1, false);
      out = conv(h1, -1, R.cout, 3, 1, 0, 2, R.c2_w, true, R.c2_b, -1, r, -1, true, -1, -1, 0, 0, R.film_off);
      drop(r);
    } else {
      out = conv(h1, -1, R.cout, 3, 1, 0, 2, R.c2_w, true, R.c2_b, -1, x0, x1, true, -1, -1, 0, 0, R.film_off);
    }
    drop(h1);
    if (R.attn) {
      fold(out, -1, R.an_w, R.an_b);
      const int qkv = conv(out, -1, 3 * R.cout, 1, 1, 0, 1, R.qkv_w, false, 0, -1, -1, -1, false);
      const int o = make(R.cout, T[out].H, T[out].W);
      Op a; a.kind = OP_ATTN;
      a.a = T[qkv].off; a.b = T[o].off; a.i0 = T[out].H * T[out].W; a.i1 = R.cout;
      ops.push_back(a);
      if (train) { Rec r; r.kind = R_ATTN; r.qkv = qkv; r.o = o; P->recs.push_back(r); }
      flops += 4.0 * B * (double)a.i0 * (double)a.i0 * R.cout;
      drop(qkv);
      const int out2 = conv(o, -1, R.cout, 1, 1, 0, 0, R.ao_w, true, R.ao_b, -1, out, -1, true);
      drop(o);
      drop(out);
      out = out2;
    }
    return out;
  }
};

// the UNet.forward walk (downs -> mid -> ups -> final), emitting ops through the builder
static void walk_forward(sr3_plan* P, Builder& bld, int cond_channels) {
  const sr3_unet_desc& d = P->d;
  std::vector<Op>& ops = bld.ops;
  const int B = bld.B;
  const int S = d.image_size, inner = d.inner_channel;

  { Op o; o.kind = OP_EMBED; ops.push_back(o); }
  bld.flops += 2.0 * B * (2.0 * 4 * inner * inner + (double)P->F * inner);

  auto tap = [&](const std::string& name, int h) {
    if (!bld.train) P->taps.push_back({name, bld.T[h].off, bld.T[h].C, bld.T[h].H, bld.T[h].W});
  };
  std::vector<int> feats;
  int cur = -1;
  for (auto& L : P->downs) {
    if (L.kind == 0) {
      cur = bld.make(L.cout, S, S);
      Op o; o.kind = OP_CONV_IN;
      o.e = bld.T[cur].off; o.p0 = L.w; o.p1 = L.b;
      o.i0 = d.in_channel - cond_channels; o.i1 = cond_channels; o.i2 = L.cout; o.i3 = S;
      if (const int slices = P->fuse_stats ? conv_in_stat_slices(d.in_channel, S, S, L.cout) : 0) {
        bld.stat_slot(cur, slices);          // the MFMA form writes the GroupNorm partials of its output itself
        o.has_ostat = true; o.f = bld.T[cur].stat_off;
      }
      ops.push_back(o);
      if (bld.train) { Rec r; r.kind = R_CONV_IN; r.out = cur; r.w = L.w; r.bias = L.b; P->recs.push_back(r); }
      bld.flops += 2.0 * B * S * S * (double)L.cout * L.cin * 9;
    } else if (L.kind == 1) {
      cur = bld.res_block(cur, -1, L.res);   // the input stays alive: it is a skip feature
    } else {
      cur = bld.conv(cur, -1, L.cout, 3, 2, 0, 0, L.w, true, L.b, -1, -1, -1, true);
    }
    feats.push_back(cur);
    tap(L.name, cur);
  }
  // `cur` is feats.back() here: skip features are released by the up block that consumes them
  bool cur_is_skip = true;
  for (auto& L : P->mid) {
    const int nxt = bld.res_block(cur, -1, L.res);
    if (!cur_is_skip) bld.drop(cur);
    cur = nxt;
    cur_is_skip = false;
    tap(L.name, cur);
  }
  for (auto& L : P->ups) {
    int nxt;
    if (L.kind == 1) {
      const int skip = feats.back();
      feats.pop_back();
      nxt = bld.res_block(cur, skip, L.res);
      bld.drop(cur);
      bld.drop(skip);
    } else {
      nxt = bld.conv(cur, -1, L.cout, 3, 1, 1, 0, L.w, true, L.b, -1, -1, -1, true);
      bld.drop(cur);
    }
    cur = nxt;
    tap(L.name, cur);
  }
  bld.fold(cur, -1, P->fin_gn_w, P->fin_gn_b);
  {
    Op o; o.kind = OP_CONV_OUT;
    o.a = bld.T[cur].off; o.p0 = P->fin_w; o.p1 = P->fin_b; o.i0 = bld.T[cur].C; o.i1 = P->out_ch; o.i2 = S;
    o.ss_rel = bld.cur_ss;
    ops.push_back(o);
    if (bld.train) {
      Rec r; r.kind = R_CONV_OUT; r.x0 = cur; r.w = P->fin_w; r.bias = P->fin_b; r.gamma = P->fin_gn_w; r.beta = P->fin_gn_b;
      r.ss_off = bld.cur_ss; r.mr_off = bld.cur_mr; r.act = 2;
      P->recs.push_back(r);
    }
    bld.flops += 2.0 * B * S * S * (double)P->out_ch * bld.T[cur].C * 9;
  }
}

// derived (Winograd) filters of every 3x3 stride-1 conv: the two convs of each ResnetBlock and the Upsample convs.  Slot of a
// conv: its fp32 fragment-major filters, followed -- plan option wino_split -- by the 3 x bf16 split form of the same filters
// (both are kept: the four-image 8x8 tile, the dropout instantiation and the training plan read the fp32 form)
void layout_derived(sr3_plan* P) {
  P->derived.clear();
  P->derived_of.clear();
  size_t dcur = 0;
  auto reg = [&](size_t w, int Cout, int Cin) {
    if (Cin & 3) return;
    P->derived.push_back({w, Cout, Cin, dcur});
    P->derived_of[w] = dcur;
    dcur += wino_weight_floats(Cout, Cin) + (P->wino_split ? wino_weight_floats(Cout, Cin, true) : 0);
  };
  for (auto* v : {&P->downs, &P->mid, &P->ups})
    for (auto& L : *v) {
      if (L.kind == 1) { reg(L.res.c1_w, L.res.cout, L.res.cin); reg(L.res.c2_w, L.res.cout, L.res.cout); }
      else if (L.kind == 3) reg(L.w, L.cout, L.cin);
    }
  // the im2col SPLIT tiles' weights, pre-split (plan option gemm_split): res_conv and the attention projections (1x1), Downsample
  // (3x3 stride 2; Cout <= 64 stays on the fp32 MFMA: Builder::conv)
  P->wsplits.clear();
  P->wsplit_of.clear();
  if (P->gemm_split && (P->gemm_wpre || P->gemm2)) {
    auto regw = [&](size_t w, int Cout, int taps, int Cin) {
      if (Cin & 3) return;
      if (!P->gemm_wpre && ((taps != 1 && !(taps == 9 && P->gemm_s2)) || (Cout & (P->gemm_n64 ? 63 : 127)) || (Cin & 31))) return;     // gemm2 alone: only what gemm1x1.hip can take (1x1; Downsample's 3x3 stride 2)
      P->wsplits.push_back({w, Cout, taps, Cin, dcur});
      P->wsplit_of[w] = dcur;
      dcur += igemm_wsplit_floats(Cout, taps, Cin);
    };
    for (auto* v : {&P->downs, &P->mid, &P->ups})
      for (auto& L : *v) {
        if (L.kind == 1) {
          if (L.res.has_rc) regw(L.res.rc_w, L.res.cout, 1, L.res.cin);
          if (L.res.attn) { regw(L.res.qkv_w, 3 * L.res.cout, 1, L.res.cout); regw(L.res.ao_w, L.res.cout, 1, L.res.cout); }
        } else if (L.kind == 2 && (L.cout > 64 || (P->gemm2 && P->gemm_s2 && P->gemm_n64 && L.cout == 64))) {
          regw(L.w, L.cout, 9, L.cin);
        }
      }
  }
  P->derived_floats = dcur;
  P->derived_from = nullptr;
  if (P->derived_bound_bytes < dcur * sizeof(float)) { P->derived_ptr = nullptr; P->derived_bound_bytes = 0; }   // re-bind a larger one
}

static int build_forward(sr3_plan* P, int B, int cond_channels) {
  if (P->built_batch == B && P->built_cond == cond_channels) return SR3_OK;
  const sr3_unet_desc& d = P->d;
  if (B <= 0) { set_error("batch must be > 0"); return SR3_E_BADARG; }
  if (cond_channels < 0 || cond_channels >= d.in_channel) { set_error("cond_channels %d out of range (in_channel %d)", cond_channels, d.in_channel); return SR3_E_BADARG; }
  P->ops.clear();
  P->taps.clear();
  Builder bld(P, B);
  const int inner = d.inner_channel;
  walk_forward(P, bld, cond_channels);
  if (P->fork_side && !P->ops.empty() && P->ops[0].kind == OP_EMBED) {
    // ... and the embedding MLP + FiLM projections (first op, reads only the noise level): beside the input conv, joined by the first conv that
    // adds a FiLM row
    for (Op& o : P->ops)
      if (o.kind == OP_CONV && o.has_film) {
        if (o.wait_id < 0) { o.wait_id = bld.n_side; P->ops[0].side_id = bld.n_side++; }
        break;
      }
  }
  // ---- fixed regions after the activation arena (high-water mark) ----
  size_t off = (bld.act.high + 255) & ~(size_t)255;
  P->stats_off = off; P->stats_bytes = bld.stats_cursor; off += (bld.stats_cursor + 255) & ~(size_t)255;
  P->ss_off = off; off += ((size_t)B * std::max(bld.max_cin, 4) * 2 * sizeof(float) + 255) & ~(size_t)255;
  P->temb_off = off; off += ((size_t)B * inner * sizeof(float) + 255) & ~(size_t)255;
  P->film_off = off; off += ((size_t)B * P->F * sizeof(float) + 255) & ~(size_t)255;
  P->scratch_off = off; P->scratch_bytes = bld.max_scratch; off += (bld.max_scratch + 255) & ~(size_t)255;
  P->ws_bytes = off;
  P->flops = bld.flops;
  P->built_batch = B;
  P->built_cond = cond_channels;
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// forward driver
// ---------------------------------------------------------------------------------------------
Regions infer_regions(const sr3_plan* P) {
  Regions r;
  r.ops = &P->ops; r.stats_off = P->stats_off; r.ss_off = P->ss_off; r.mr_off = 0; r.temb_off = P->temb_off;
  r.film_off = P->film_off; r.scratch_off = P->scratch_off; r.scratch_bytes = P->scratch_bytes;
  return r;
}

static FoldTail make_fold_tail(const Op& o, int groups, const float* params, char* ws, const Regions& R) {
  FoldTail f;
  memset(&f, 0, sizeof(f));
  f.groups = groups; f.Ctot = o.f_Ctot; f.c_off = o.f_coff;
  if (o.f_has_o) { f.ostat = reinterpret_cast<const double*>(ws + R.stats_off + o.f_ostat); f.oC = o.f_oC; f.oT = o.f_oT; f.o_off = o.f_ooff; }
  f.gamma = params + o.f_gamma; f.beta = params + o.f_beta; f.eps = 1e-5f;
  f.ss = reinterpret_cast<float*>(ws + R.ss_off + o.f_ss_rel);
  f.mr = o.f_has_mr ? reinterpret_cast<float*>(ws + R.mr_off + o.f_mr_rel) : nullptr;
  return f;
}

int run_forward(sr3_plan* P, const Regions& R, const float* x, const float* cond, int cond_channels, const float* level,
                const int64_t* tstep, const float* freq, const float* level_table, const int* step_dev,
                const float* params, char* ws, float* eps_out, int B, hipStream_t st,
                hipEvent_t* ev, hipEvent_t* mid, const DropCfg* drop, const StepFuse* fuse) {
  const sr3_unet_desc& d = P->d;
  size_t op_index = 0;
  float* film = reinterpret_cast<float*>(ws + R.film_off);
  // plan option fork_side: ops marked side_id run on the plan's side stream between a fork event (recorded on the caller's stream where the
  // op sits in the list) and a join event their consumer (wait_id) waits for; under per-op timing (ev) everything stays on one stream
  const bool forking = !ev && R.ops == &P->ops;
  hipStream_t const main_st = st;
  for (const Op& o : *R.ops) {
    int rc = SR3_OK;
    st = main_st;
    if (ev) SR3_HIP(hipEventRecord(ev[op_index], st));
    ++op_index;
    if (forking && o.wait_id >= 0 && o.wait_id < (int)P->join_ev.size())
      SR3_HIP(hipStreamWaitEvent(main_st, P->join_ev[o.wait_id], 0));
    if (forking && o.side_id >= 0) {
      if (!P->side_stream) SR3_HIP(hipStreamCreateWithFlags(&P->side_stream, hipStreamNonBlocking));
      while ((int)P->fork_ev.size() <= o.side_id) {
        hipEvent_t a, b;
        SR3_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        SR3_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        P->fork_ev.push_back(a); P->join_ev.push_back(b);
      }
      SR3_HIP(hipEventRecord(P->fork_ev[o.side_id], main_st));
      SR3_HIP(hipStreamWaitEvent(P->side_stream, P->fork_ev[o.side_id], 0));
      st = P->side_stream;
    }
    switch (o.kind) {
      case OP_RESERVED:
        break;
      case OP_EMBED: {
        EmbedParams e;
        memset(&e, 0, sizeof(e));
        e.variant = d.variant; e.B = B; e.inner = d.inner_channel;
        e.level = level; e.tstep = tstep; e.level_table = level_table; e.step_dev = step_dev; e.freq = freq;
        e.step_out = fuse ? const_cast<int*>(fuse->step_cur) : nullptr;
        e.w1 = params + P->emb_w1; e.b1 = params + P->emb_b1; e.w2 = params + P->emb_w2; e.b2 = params + P->emb_b2;
        e.wf = params + P->film_w; e.bf = params + P->film_b; e.F = P->F;
        e.temb = reinterpret_cast<float*>(ws + R.temb_off); e.film = film;
        rc = embed_forward(e, st);
        break;
      }
      case OP_CONV_IN: {
        // virtual concat order is [cond | x] (diffusion.py:157); unconditional: x only
        const float* a = cond_channels > 0 ? cond : x;
        const int Ca = cond_channels > 0 ? cond_channels : o.i0;
        const float* b = cond_channels > 0 ? x : nullptr;
        const int Cb = cond_channels > 0 ? o.i0 : 0;
        rc = conv_in_nchw(a, Ca, b, Cb, B, o.i3, o.i3, params + o.p0, params + o.p1, o.i2,
                          reinterpret_cast<float*>(ws + o.e),
                          o.has_ostat ? reinterpret_cast<double*>(ws + R.stats_off + o.f) : nullptr, st);
        break;
      }
      case OP_STATS:
        if (o.fold_fused) {
          FoldTail ft = make_fold_tail(o, d.norm_groups, params, ws, R);
          rc = chan_stats_fold(reinterpret_cast<const float*>(ws + o.a), B, o.i0, o.i1, reinterpret_cast<double*>(ws + R.stats_off + o.b), ft, st);
          break;
        }
        rc = chan_stats(reinterpret_cast<const float*>(ws + o.a), B, o.i0, o.i1,
                        reinterpret_cast<double*>(ws + R.stats_off + o.b), st);
        break;
      case OP_FOLD:
        rc = gn_finalize(reinterpret_cast<const double*>(ws + R.stats_off + o.a), o.i0, o.i3,
                         o.has_st1 ? reinterpret_cast<const double*>(ws + R.stats_off + o.b) : nullptr,
                         o.has_st1 ? o.i1 : 0, o.has_st1 ? o.i4 : 0, B, o.i2, d.norm_groups, params + o.p0,
                         params + o.p1, 1e-5f, reinterpret_cast<float*>(ws + R.ss_off + o.ss_rel), st,
                         o.has_mr ? reinterpret_cast<float*>(ws + R.mr_off + o.mr_rel) : nullptr);
        break;
      case OP_CONV: {
        ConvParams c = o.cp;
        c.src0 = reinterpret_cast<const float*>(ws + o.a);
        c.src1 = o.has_src1 ? reinterpret_cast<const float*>(ws + o.b) : nullptr;
        c.w = params + o.p0;
        c.bias = o.has_bias ? params + o.p1 : nullptr;
        c.ss = c.act ? reinterpret_cast<const float*>(ws + R.ss_off + o.ss_rel) : nullptr;
        c.film = o.has_film ? film + o.i0 : nullptr;
        c.res0 = o.has_res ? reinterpret_cast<const float*>(ws + o.c) : nullptr;
        c.res1 = o.has_res1 ? reinterpret_cast<const float*>(ws + o.d) : nullptr;
        c.out = reinterpret_cast<float*>(ws + o.e);
        c.ostat = o.has_ostat ? reinterpret_cast<double*>(ws + R.stats_off + o.f) : nullptr;
        if (drop && o.has_drop && drop->thresh != 0) {
          c.drop_seed = drop_layer_seed(drop->seed, o.drop_key); c.drop_thresh = drop->thresh; c.drop_scale = drop->scale;
        }
        if (o.has_x2) {
          c.x2_src0 = reinterpret_cast<const float*>(ws + o.g);
          c.x2_src1 = o.has_x21 ? reinterpret_cast<const float*>(ws + o.h) : nullptr;
          c.x2_w = params + o.p2;
          c.x2_bias = params + o.p3;
        }
        if (o.tile_cfg == 11) {
          if (!P->derived_ptr) { set_error("the plan's derived (Winograd) weights are not bound: call sr3_plan_bind_derived + sr3_plan_prepare_derived"); return SR3_E_BADARG; }
          // stale filters must fail loudly, not compute with the previous weights: the buffer has to have been prepared from
          // THIS arena, under the current options, and not invalidated since (sr3_plan_invalidate_derived after an optimizer step)
          if (P->derived_from != params) {
            set_error("the plan's derived (Winograd) weights are stale or were prepared from another arena: call sr3_plan_prepare_derived");
            return SR3_E_BADARG;
          }
          c.wino_u = P->derived_ptr + o.wino_off;
        }
        if (o.has_wsplit && ((o.tile_cfg >= 1 && o.tile_cfg <= 4) || o.tile_cfg == 22) && c.igemm_split) {
          if (!P->derived_ptr || P->derived_from != params) {
            set_error("the plan's derived (pre-split 1x1 / stride-2) weights are not bound or stale: call sr3_plan_bind_derived + sr3_plan_prepare_derived");
            return SR3_E_BADARG;
          }
          c.w_split = P->derived_ptr + o.wsplit_off;
        }
        if (mid && o.ksplit > 1) conv_set_mid_event(mid[op_index - 1]);
        FoldTail ft;
        if (o.fold_fused) { ft = make_fold_tail(o, d.norm_groups, params, ws, R); c.fold = &ft; }
        rc = conv_forward(c, o.tile_cfg, o.ksplit, reinterpret_cast<float*>(ws + R.scratch_off), R.scratch_bytes, st);
        break;
      }
      case OP_ATTN:
        rc = attention_forward(reinterpret_cast<const float*>(ws + o.a), B, o.i0, o.i1,
                               reinterpret_cast<float*>(ws + o.b), st, P->attn_split);
        break;
      case OP_CONV_OUT:
        rc = conv_out_nchw(reinterpret_cast<const float*>(ws + o.a), reinterpret_cast<const float*>(ws + R.ss_off + o.ss_rel),
                           B, o.i2, o.i2, o.i0, params + o.p0, params + o.p1, o.i1, eps_out, st, fuse);
        break;
    }
    if (rc) return rc;
    if (forking && o.side_id >= 0) SR3_HIP(hipEventRecord(P->join_ev[o.side_id], st));
  }
  st = main_st;
  if (ev) SR3_HIP(hipEventRecord(ev[op_index], st));
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// training plan: the same forward walk with every activation kept and persistent GroupNorm tables,
// a gradient mirror of the activation arena, and the scratch the backward walk needs
// ---------------------------------------------------------------------------------------------
int build_train(sr3_plan* P, int B, int cond_channels) {
  if (P->train_batch == B && P->train_cond == cond_channels) return SR3_OK;
  const sr3_unet_desc& d = P->d;
  if (B <= 0) { set_error("batch must be > 0"); return SR3_E_BADARG; }
  if (cond_channels < 0 || cond_channels >= d.in_channel) { set_error("cond_channels out of range"); return SR3_E_BADARG; }
  P->tops.clear();
  P->recs.clear();
  Builder bld(P, B, true);
  walk_forward(P, bld, cond_channels);
  P->ttens = bld.T;
  const int S = d.image_size, inner = d.inner_channel, G = d.norm_groups;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = al(bld.act.high);
  P->t_act_bytes = off;
  off *= 2;                                   // gradients mirror the activations at +t_act_bytes
  P->t_stats_off = off; off += al(bld.stats_cursor);
  P->t_gn_off = off; off += al(bld.gn_cursor);
  P->t_misc_off = off; off += al(bld.mr_cursor);           // mean / rstd tables
  P->t_temb_off = off; off += al((size_t)B * inner * sizeof(float));
  P->t_film_off = off; off += al((size_t)B * P->F * sizeof(float));
  P->t_scratch_off = off; P->t_scratch_bytes = bld.max_scratch; off += al(bld.max_scratch);
  // backward scratch
  size_t max_dA = bld.max_dA, max_wt = bld.max_wt, max_slab = 0, max_part = 0, max_dwtmp = 0, max_wu = 0;
  size_t max_bscratch = 0;                    // split-K slabs of the data-gradient convs
  for (const Rec& r : P->recs) {
    ConvParams c;
    memset(&c, 0, sizeof(c));
    if (r.kind == R_CONV) {
      const Tensor& x0 = P->ttens[r.x0];
      const Tensor& o = P->ttens[r.out];
      c.C0 = x0.C; c.C1 = r.x1 >= 0 ? P->ttens[r.x1].C : 0; c.B = B; c.Hs = x0.H; c.Ws = x0.W; c.ups = r.ups;
      c.stride = r.stride; c.ksize = r.ksize; c.Ho = o.H; c.Wo = o.W; c.Cout = o.C;
      if (r.act) { c.C0 = c.C0 + c.C1; c.C1 = 0; }     // run time: the activated input is materialised (single source)
      max_slab = std::max(max_slab, wgrad_slab_bytes(c, nullptr));
      max_part = std::max(max_part, act_bwd_part_bytes(B, x0.H * x0.W, c.C0 + c.C1));
      max_part = std::max(max_part, (size_t)B * chan_stats_slices(B, o.H * o.W, o.C) * o.C * 2 * sizeof(double));
      // dgrad conv: src = dOut (or its zero-inserted version), Cout' = Cin
      ConvParams g;
      memset(&g, 0, sizeof(g));
      g.C0 = o.C; g.B = B; g.Hs = x0.H << r.ups; g.Ws = x0.W << r.ups; g.stride = 1; g.ksize = r.ksize;
      g.Ho = g.Hs; g.Wo = g.Ws; g.Cout = c.C0 + c.C1;
      g.igemm_split = (P->gemm_split && !(g.ksize == 3 && g.Cout <= 64)) ? 1 : 0;      // (what dgrad_conv launches with: train_plan.hip)
      max_bscratch = std::max(max_bscratch, conv_splitk_bytes(g, 0, 0));
      {   // the data gradient of a 3x3 conv runs on the Winograd kernel where it fits: its slabs and transformed filters
        WinoGeom wg;
        if (P->winograd && r.ksize == 3 && wino_geometry(g, &wg)) {
          g.wino_split = (P->wino_split && wg.NB == 1) ? (P->wino2 ? 2 : 1) : 0;      // (what dgrad_conv launches with: its split-K choice depends on it)
          max_bscratch = std::max(max_bscratch, conv_splitk_bytes(g, 11, 0));
          max_wu = std::max(max_wu, wino_weight_floats(g.Cout, g.C0, P->wino_split && wg.NB == 1) * sizeof(float));
        }
      }
      if (r.has_q) {
        ConvParams q;
        memset(&q, 0, sizeof(q));
        q.C0 = P->ttens[r.q0].C; q.C1 = r.q1 >= 0 ? P->ttens[r.q1].C : 0; q.B = B; q.Hs = o.H; q.Ws = o.W; q.stride = 1;
        q.ksize = 1; q.Ho = o.H; q.Wo = o.W; q.Cout = o.C;
        max_slab = std::max(max_slab, wgrad_slab_bytes(q, nullptr));
        ConvParams gq;
        memset(&gq, 0, sizeof(gq));
        gq.C0 = o.C; gq.B = B; gq.Hs = o.H; gq.Ws = o.W; gq.stride = 1; gq.ksize = 1; gq.Ho = o.H; gq.Wo = o.W;
        gq.Cout = q.C0 + q.C1;
        max_bscratch = std::max(max_bscratch, conv_splitk_bytes(gq, 0, 0));
      }
    } else if (r.kind == R_CONV_IN) {
      c.C0 = 8; c.B = B; c.Hs = S; c.Ws = S; c.stride = 1; c.ksize = 3; c.Ho = S; c.Wo = S; c.Cout = P->ttens[r.out].C;
      max_slab = std::max(max_slab, wgrad_slab_bytes(c, nullptr));
      max_dwtmp = std::max(max_dwtmp, (size_t)c.Cout * 9 * 8 * sizeof(float));
      max_part = std::max(max_part, (size_t)B * chan_stats_slices(B, S * S, c.Cout) * c.Cout * 2 * sizeof(double));
    } else if (r.kind == R_CONV_OUT) {
      const Tensor& x0 = P->ttens[r.x0];
      c.C0 = x0.C; c.B = B; c.Hs = S; c.Ws = S; c.stride = 1; c.ksize = 3; c.Ho = S; c.Wo = S; c.Cout = 4;
      max_slab = std::max(max_slab, wgrad_slab_bytes(c, nullptr));
      max_dwtmp = std::max(max_dwtmp, (size_t)4 * 9 * x0.C * sizeof(float));
      max_dA = std::max(max_dA, (size_t)B * S * S * x0.C * sizeof(float));
      max_wt = std::max(max_wt, (size_t)x0.C * 9 * 4 * sizeof(float));
      max_part = std::max(max_part, act_bwd_part_bytes(B, S * S, x0.C));
      max_part = std::max(max_part, (size_t)B * chan_stats_slices(B, S * S, 4) * 4 * 2 * sizeof(double));
      ConvParams g;
      memset(&g, 0, sizeof(g));
      g.C0 = 4; g.B = B; g.Hs = S; g.Ws = S; g.stride = 1; g.ksize = 3; g.Ho = S; g.Wo = S; g.Cout = x0.C;
      max_bscratch = std::max(max_bscratch, conv_splitk_bytes(g, 0, 0));
    }
  }
  for (const Rec& r : P->recs)                // ... and the dK / dV slabs of the attention backward (attention_bwd.hip)
    if (r.kind == R_ATTN) {
      const Tensor& o = P->ttens[r.o];
      max_bscratch = std::max(max_bscratch, attention_backward_scratch_bytes(B, o.H * o.W, o.C));
    }
  if (max_bscratch > P->t_scratch_bytes) {    // the forward's split-K region doubles as the backward's
    off -= al(P->t_scratch_bytes);
    P->t_scratch_bytes = max_bscratch;
    off += al(max_bscratch);
  }
  P->t_dA_off = off; off += al(max_dA);
  P->t_a_off = off; off += al(max_dA);       // materialised activated input of the weight-gradient GEMM
  P->t_z_off = off; off += al(bld.max_z);
  P->t_dq_off = off; off += al(bld.max_dq);
  P->t_wt_off = off; off += al(max_wt);
  P->t_wu_off = off; P->t_wu_bytes = max_wu; off += al(max_wu);
  P->t_slab_off = off; off += al(max_slab);
  P->t_part_off = off; off += al(max_part);
  P->t_gs_off = off; off += al((size_t)B * G * 2 * sizeof(double));
  P->t_dfilm_off = off; off += al((size_t)B * P->F * sizeof(float));
  P->t_xnoisy_off = off; off += al((size_t)B * (d.in_channel - cond_channels) * S * S * sizeof(float));
  P->t_eps_off = off; off += al((size_t)B * P->out_ch * S * S * sizeof(float));
  P->t_geps_off = off; off += al((size_t)B * S * S * 4 * sizeof(float));
  P->t_inpad_off = off; off += al((size_t)B * S * S * 8 * sizeof(float));
  P->t_dwtmp_off = off; off += al(4096 * sizeof(double)) + al(max_dwtmp);      // [loss partials | dw temp]
  P->t_embscr_off = off; off += al((size_t)B * (13 + 16) * inner * sizeof(float));     // (+ the 16 row chunks of k_film_bwd_input)
  // gradient-ready marks: t_unproc_max[k] = largest arena offset (exclusive end) among the parameters whose
  // gradients are still unwritten once records k .. end have been processed (records < k + the FiLM /
  // embedding block at the arena head, which is written last)
  {
    const size_t nrec = P->recs.size();
    P->t_unproc_max.assign(nrec + 1, 0);
    size_t head_end = P->emb_b2 + (size_t)inner;
    head_end = std::max(head_end, P->film_b + (size_t)P->F);
    size_t run = head_end;
    auto pend = [&](size_t off, size_t n) { return off + n; };
    for (size_t k = 0; k < nrec; ++k) {
      P->t_unproc_max[k] = run;
      const Rec& r = P->recs[k];
      size_t e = 0;
      if (r.kind == R_ATTN) { /* no parameters */ }
      else {
        const int cout = r.kind == R_CONV_OUT ? P->out_ch : P->ttens[r.out >= 0 ? r.out : 0].C;
        size_t cin = 0;
        if (r.kind == R_CONV_IN) cin = d.in_channel;
        else cin = (size_t)P->ttens[r.x0].C + (r.x1 >= 0 ? P->ttens[r.x1].C : 0);
        e = std::max(e, pend(r.w, (size_t)cout * r.ksize * r.ksize * cin));
        e = std::max(e, pend(r.bias, (size_t)cout));
        if (r.act) { e = std::max(e, pend(r.gamma, cin)); e = std::max(e, pend(r.beta, cin)); }
        if (r.has_q) {
          const size_t cq = (size_t)P->ttens[r.q0].C + (r.q1 >= 0 ? P->ttens[r.q1].C : 0);
          e = std::max(e, pend(r.qw, (size_t)cout * cq));
          e = std::max(e, pend(r.qb, (size_t)cout));
        }
      }
      run = std::max(run, e);
    }
    P->t_unproc_max[nrec] = run;
  }
  P->t_ws_bytes = off;
  P->train_batch = B;
  P->train_cond = cond_channels;
  return SR3_OK;
}

}  // namespace sr3

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int sr3_version(void) { return SR3_ABI_VERSION; }
const char* sr3_last_error(void) { return sr3::last_error(); }
int sr3_selftest_split3(int* scratch_dev, int* mismatches, void* stream) {
  if (!scratch_dev || !mismatches) { sr3::set_error("null argument"); return SR3_E_BADARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = sr3::split3_selftest(scratch_dev, st);
  if (rc) return rc;
  SR3_HIP(hipMemcpyAsync(mismatches, scratch_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  SR3_HIP(hipStreamSynchronize(st));
  if (*mismatches != 0) { sr3::set_error("split3 self-test: %d of 2097152 elements do not satisfy x == h + m + l exactly (compiler / ISA change?)", *mismatches); return SR3_E_UNSUPPORTED; }
  return SR3_OK;
}

int sr3_plan_create(const sr3_unet_desc* desc, sr3_plan** out) {
  if (!desc || !out) { set_error("null argument"); return SR3_E_BADARG; }
  sr3_plan* P = new (std::nothrow) sr3_plan();
  if (!P) { set_error("out of host memory"); return SR3_E_NOMEM; }
#ifdef SR3_EXPERIMENTS
  // A/B builds only: defaults of plan options from the environment (a release library's arithmetic does not depend on the environment)
  { const char* e = getenv("SR3_WGRAD_SPLIT"); if (e) P->wgrad_split = atoi(e); }
  { const char* e = getenv("SR3_WINO2"); if (e) P->wino2 = atoi(e); }
#endif
  P->d = *desc;
  const int rc = build_structure(P);
  if (rc) { delete P; *out = nullptr; return rc; }
  *out = P;
  return SR3_OK;
}
void sr3_plan_destroy(sr3_plan* plan) { delete plan; }
int sr3_plan_num_params(const sr3_plan* plan) { return plan ? (int)plan->params.size() : 0; }
int sr3_plan_param_info(const sr3_plan* plan, int index, sr3_param_info* out) {
  if (!plan || !out || index < 0 || index >= (int)plan->params.size()) { set_error("bad param index"); return SR3_E_BADARG; }
  *out = plan->params[index];
  return SR3_OK;
}
size_t sr3_plan_param_floats(const sr3_plan* plan) { return plan ? plan->param_floats : 0; }
int sr3_plan_num_ops(sr3_plan* plan, int batch) {
  if (!plan) return 0;
  const int cond = plan->built_cond >= 0 ? plan->built_cond : 0;
  if (build_forward(plan, batch, cond)) return -1;
  return (int)plan->ops.size();
}
int sr3_plan_op_info(sr3_plan* plan, int batch, int index, sr3_op_info* out) {
  if (!plan || !out) { set_error("null argument"); return SR3_E_BADARG; }
  const int cond = plan->built_cond >= 0 ? plan->built_cond : 0;
  const int rc = build_forward(plan, batch, cond);
  if (rc) return rc;
  if (index < 0 || index >= (int)plan->ops.size()) { set_error("op index out of range"); return SR3_E_BADARG; }
  const Op& o = plan->ops[index];
  memset(out, 0, sizeof(*out));
  out->kind = (int)o.kind * 10;
  if (o.kind == OP_CONV) {
    const ConvParams& c = o.cp;
    out->tile_cfg = (o.tile_cfg == 11 && o.cp.wino_split) ? 11 + o.cp.wino_split
                    : (o.tile_cfg >= 1 && o.tile_cfg <= 4 && o.cp.igemm_split) ? (o.has_wsplit ? 17 : 13) + o.tile_cfg : o.tile_cfg;
    out->ksplit = o.ksplit;
    out->ksize = c.ksize; out->stride = c.stride; out->upsample = c.ups;
    out->cin = c.C0 + c.C1; out->cout = c.Cout; out->h_out = c.Ho; out->w_out = c.Wo;
    out->fused_res_conv_cin = o.has_x2 ? c.x2_C0 + c.x2_C1 : 0;
    out->fused_output_stats = o.has_ostat ? 1 : 0;
    out->flops = 2.0 * c.B * c.Ho * c.Wo * (double)c.Cout * ((double)out->cin * c.ksize * c.ksize + out->fused_res_conv_cin);
  } else if (o.kind == OP_CONV_IN) {
    out->ksize = 3; out->stride = 1; out->cin = o.i0 + o.i1; out->cout = o.i2; out->h_out = out->w_out = o.i3;
    out->fused_output_stats = o.has_ostat ? 1 : 0;
  } else if (o.kind == OP_ATTN) {
    out->h_out = o.i0; out->cin = out->cout = o.i1;          // tokens, channels
    out->flops = 4.0 * batch * (double)o.i0 * (double)o.i0 * o.i1;
  }
  return SR3_OK;
}
int sr3_plan_op_side(sr3_plan* plan, int batch, int index, int* side_id, int* wait_id) {
  if (!plan) { set_error("null argument"); return SR3_E_BADARG; }
  const int cond = plan->built_cond >= 0 ? plan->built_cond : 0;
  const int rc = build_forward(plan, batch, cond);
  if (rc) return rc;
  if (index < 0 || index >= (int)plan->ops.size()) { set_error("op index out of range"); return SR3_E_BADARG; }
  if (side_id) *side_id = plan->ops[index].side_id;
  if (wait_id) *wait_id = plan->ops[index].wait_id;
  return SR3_OK;
}
double sr3_plan_forward_flops(sr3_plan* plan, int batch) {
  if (!plan) return 0;
  const int cond = plan->built_cond >= 0 ? plan->built_cond : 0;
  if (build_forward(plan, batch, cond)) return -1;
  return plan->flops;
}
int sr3_plan_set_option(sr3_plan* plan, const char* key, int value) {
  if (!plan || !key) return SR3_E_BADARG;
  int* slot = nullptr;
  if (!strcmp(key, "fuse_stats")) slot = &plan->fuse_stats;
  else if (!strcmp(key, "tile_cfg")) slot = &plan->tile_cfg;
  else if (!strcmp(key, "ksplit")) slot = &plan->ksplit;
  else if (!strcmp(key, "keep_all")) slot = &plan->keep_all;
  else if (!strcmp(key, "fuse_res")) slot = &plan->fuse_res;
  else if (!strcmp(key, "split_bf16")) slot = &plan->split_bf16;
  else if (!strcmp(key, "winograd")) slot = &plan->winograd;
  else if (!strcmp(key, "wino_split")) slot = &plan->wino_split;
  else if (!strcmp(key, "wino_split8")) slot = &plan->wino_split8;
  else if (!strcmp(key, "wgrad_split")) { const int prev = plan->wgrad_split; plan->wgrad_split = value; return prev; }   // no rebuild (the slabs are sized for both)
  else if (!strcmp(key, "attn_split")) { const int prev = plan->attn_split; plan->attn_split = value; return prev; }   // no rebuild
  else if (!strcmp(key, "gemm_split")) slot = &plan->gemm_split;
  else if (!strcmp(key, "gemm_wpre")) slot = &plan->gemm_wpre;
  else if (!strcmp(key, "gemm2")) slot = &plan->gemm2;
  else if (!strcmp(key, "gemm_s2")) slot = &plan->gemm_s2;
  else if (!strcmp(key, "fork_side")) slot = &plan->fork_side;
  else if (!strcmp(key, "gemm_n64")) slot = &plan->gemm_n64;
  else if (!strcmp(key, "fold_fuse")) slot = &plan->fold_fuse;
  else if (!strcmp(key, "gemm_tile")) slot = &plan->gemm_tile;
  else if (!strcmp(key, "wino2")) slot = &plan->wino2;
  else if (!strcmp(key, "loss_l2")) { const int prev = plan->loss_l2; plan->loss_l2 = value; return prev; }   // no rebuild
  if (!slot) { set_error("unknown option %s", key); return SR3_E_BADARG; }
#ifndef SR3_EXPERIMENTS
  if (slot == &plan->split_bf16 && value != 0) {
    set_error("option %s selects an experiment kernel that this library was built without (csrc/build.sh -DSR3_EXPERIMENTS)", key);
    return SR3_E_UNSUPPORTED;
  }
#endif
  const int prev = *slot;
  *slot = value;
  plan->built_batch = -1;
  plan->train_batch = -1;
  // which convs read transformed filters depends on these: a forward must not run on filters prepared for another choice
  if (slot == &plan->winograd || slot == &plan->tile_cfg || slot == &plan->split_bf16) plan->derived_from = nullptr;
  if ((slot == &plan->wino_split || slot == &plan->gemm_split || slot == &plan->gemm_wpre || slot == &plan->gemm2 || slot == &plan->gemm_s2 || slot == &plan->gemm_n64) && prev != value) layout_derived(plan);     // (the buffer has to be re-bound and re-prepared)
  return prev;
}
int sr3_plan_num_taps(sr3_plan* plan) { return plan ? (int)plan->taps.size() : 0; }
int sr3_plan_tap_info(sr3_plan* plan, int index, char* name, int name_len, size_t* offset, int* C, int* H, int* W) {
  if (!plan || index < 0 || index >= (int)plan->taps.size()) { set_error("bad tap index"); return SR3_E_BADARG; }
  const Tap& t = plan->taps[index];
  if (name && name_len > 0) snprintf(name, name_len, "%s", t.name.c_str());
  if (offset) *offset = t.off;
  if (C) *C = t.C;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return SR3_OK;
}

size_t sr3_plan_derived_bytes(const sr3_plan* plan) { return plan ? plan->derived_floats * sizeof(float) : 0; }
int sr3_plan_bind_derived(sr3_plan* plan, void* buffer, size_t bytes) {
  if (!plan) { set_error("null plan"); return SR3_E_BADARG; }
  if (buffer && (bytes < plan->derived_floats * sizeof(float) || ((uintptr_t)buffer & 15))) {
    set_error("derived buffer too small or misaligned (%zu < %zu)", bytes, plan->derived_floats * sizeof(float));
    return SR3_E_NOMEM;
  }
  plan->derived_ptr = static_cast<float*>(buffer);
  plan->derived_bound_bytes = buffer ? bytes : 0;
  plan->derived_from = nullptr;               // a freshly bound buffer holds nothing yet
  return SR3_OK;
}
int sr3_plan_invalidate_derived(sr3_plan* plan) {
  if (!plan) { set_error("null plan"); return SR3_E_BADARG; }
  plan->derived_from = nullptr;
  return SR3_OK;
}
int sr3_plan_prepare_derived(sr3_plan* plan, const float* params, void* stream) {
  if (!plan || !params) { set_error("null argument"); return SR3_E_BADARG; }
  if (!plan->derived_ptr) { set_error("no derived buffer bound"); return SR3_E_BADARG; }
  for (const auto& d : plan->derived) {
    int rc = wino_transform_weights(params + d.w, d.Cout, d.Cin, plan->derived_ptr + d.off, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    if (plan->wino_split) {
      rc = wino_transform_weights(params + d.w, d.Cout, d.Cin, plan->derived_ptr + d.off + wino_weight_floats(d.Cout, d.Cin),
                                  static_cast<hipStream_t>(stream), true);
      if (rc) return rc;
    }
  }
  for (const auto& d : plan->wsplits) {
    const int rc = igemm_split_weights(params + d.w, d.Cout, d.taps, d.Cin, plan->derived_ptr + d.off, static_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  plan->derived_from = params;
  return SR3_OK;
}

size_t sr3_workspace_bytes(sr3_plan* plan, int batch) {
  if (!plan) return 0;
  const int cond = plan->built_cond >= 0 ? plan->built_cond : 0;
  if (build_forward(plan, batch, cond)) return 0;
  return plan->ws_bytes;
}

int sr3_unet_forward(sr3_plan* plan, const float* x_nchw, const float* cond_nchw, int cond_channels,
                     const float* noise_level, const int64_t* timestep, const float* freq, const float* level_table,
                     const int* step_dev, const float* params, void* workspace, size_t workspace_bytes,
                     float* eps_out_nchw, int batch, void* stream) {
  if (!plan || !x_nchw || !params || !workspace || !eps_out_nchw || !freq) { set_error("null argument"); return SR3_E_BADARG; }
  if (!cond_nchw) cond_channels = 0;
  const int rc = build_forward(plan, batch, cond_channels);
  if (rc) return rc;
  if (workspace_bytes < plan->ws_bytes) { set_error("workspace too small: %zu < %zu", workspace_bytes, plan->ws_bytes); return SR3_E_NOMEM; }
  if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)x_nchw & 15) || ((uintptr_t)eps_out_nchw & 15)) {
    set_error("misaligned pointer (workspace 256 B, tensors 16 B)");
    return SR3_E_ALIGN;
  }
  if (plan->d.variant == SR3_VARIANT_SR3 && !noise_level && !step_dev) { set_error("SR3 variant needs noise_level or step_dev"); return SR3_E_BADARG; }
  if (plan->d.variant == SR3_VARIANT_DDPM && !timestep && !step_dev) { set_error("DDPM variant needs timestep or step_dev"); return SR3_E_BADARG; }
  return run_forward(plan, infer_regions(plan), x_nchw, cond_nchw, cond_channels, noise_level, timestep, freq, level_table,
                     step_dev, params, static_cast<char*>(workspace), eps_out_nchw, batch, static_cast<hipStream_t>(stream),
                     nullptr, nullptr);
}

// One whole reverse step (include/sr3_mi355x.h): the forward above with the p_sample update and the counter decrement inside the
// output conv's kernel -- two graph nodes fewer per step than sr3_unet_forward + sr3_p_sample_step + sr3_step_decrement.
int sr3_reverse_step(sr3_plan* plan, float* x_nchw, const float* cond_nchw, int cond_channels, const float* freq,
                     const float* level_table, int* step2_dev, const float* params, void* workspace, size_t workspace_bytes,
                     const float* z_nchw, const float* ta, const float* tb, const float* tc1, const float* tc2, const float* tsig,
                     int clip_denoised, float* eps_out_nchw, int batch, void* stream) {
  if (!plan || !x_nchw || !params || !workspace || !freq || !step2_dev || !ta || !tb || !tc1 || !tc2 || !tsig) { set_error("null argument"); return SR3_E_BADARG; }
  if (!cond_nchw) cond_channels = 0;
  const int rc = build_forward(plan, batch, cond_channels);
  if (rc) return rc;
  if (workspace_bytes < plan->ws_bytes) { set_error("workspace too small: %zu < %zu", workspace_bytes, plan->ws_bytes); return SR3_E_NOMEM; }
  if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)x_nchw & 15) || ((uintptr_t)eps_out_nchw & 15) || ((uintptr_t)z_nchw & 15)) {
    set_error("misaligned pointer (workspace 256 B, tensors 16 B)");
    return SR3_E_ALIGN;
  }
  if (plan->d.variant == SR3_VARIANT_SR3 && !level_table) { set_error("SR3 variant needs level_table"); return SR3_E_BADARG; }
  StepFuse f;
  f.x = x_nchw; f.z = z_nchw; f.tb = StepTables{ta, tb, tc1, tc2, tsig};
  f.step_cur = step2_dev; f.step_next = step2_dev + 1; f.clip = clip_denoised;
  // the embedding kernel reads t from slot 1 and copies it to slot 0; the tail reads slot 0 and writes t - 1 to slot 1: no kernel
  // both reads and writes a slot, so no launch of the step races with another block of itself
  return run_forward(plan, infer_regions(plan), x_nchw, cond_nchw, cond_channels, nullptr, nullptr, freq, level_table,
                     step2_dev + 1, params, static_cast<char*>(workspace), eps_out_nchw, batch, static_cast<hipStream_t>(stream),
                     nullptr, nullptr, nullptr, &f);
}

int sr3_unet_forward_profile(sr3_plan* plan, const float* x_nchw, const float* cond_nchw, int cond_channels,
                             const float* noise_level, const int64_t* timestep, const float* freq, const float* params,
                             void* workspace, size_t workspace_bytes, float* eps_out_nchw, int batch, void* stream,
                             int max_ops, float* op_ms, int* op_kind, double* op_flops, int* n_ops) {
  if (!plan || !op_ms || !op_kind || !op_flops || !n_ops) { set_error("null argument"); return SR3_E_BADARG; }
  if (!cond_nchw) cond_channels = 0;
  int rc = build_forward(plan, batch, cond_channels);
  if (rc) return rc;
  if (workspace_bytes < plan->ws_bytes) { set_error("workspace too small"); return SR3_E_NOMEM; }
  const int n = (int)plan->ops.size();
  std::vector<hipEvent_t> ev(n + 1, nullptr), mid(n, nullptr);
  for (auto* v : {&ev, &mid})
    for (auto& e : *v)
      if (hipError_t err = hipEventCreate(&e); err != hipSuccess && !rc) rc = hip_fail(err, "hipEventCreate");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!rc)
    rc = run_forward(plan, infer_regions(plan), x_nchw, cond_nchw, cond_channels, noise_level, timestep, freq, nullptr,
                     nullptr, params, static_cast<char*>(workspace), eps_out_nchw, batch, st, ev.data(), mid.data());
  if (!rc) {
    hipError_t e = hipEventSynchronize(ev[n]);
    if (e != hipSuccess) rc = hip_fail(e, "hipEventSynchronize");
  }
  if (!rc) {
    int w = 0;
    for (int i = 0; i < n && !rc; ++i) {
      float ms = 0.f;
      if (hipError_t e = hipEventElapsedTime(&ms, ev[i], ev[i + 1]); e != hipSuccess) { rc = hip_fail(e, "hipEventElapsedTime"); break; }
      const Op& o = plan->ops[i];
      int kind = (int)o.kind * 10;
      double fl = 0.0;
      float red_ms = -1.f;
      if (o.kind == OP_CONV) {
        // 51-54 im2col kernel tile configs; 55/56 halo-tile 3x3 kernel (57/58: with the fused 1x1 segment)
        //        155-158: the same four on the opt-in split-bf16 instantiations; 255/257: the 8-wave 256x128 tile
        //        (cfg 9), 355/357: its split-bf16 twin (cfg 10); 455: the Winograd F(2x2,3x3) kernel (cfg 11), 465: its four-image
        //        tile of the 8x8 maps, 555: its 3 x bf16 split instantiation (plan option wino_split)
        {
          static const int base[13] = {0, 1, 2, 3, 4, 5, 6, 105, 106, 205, 305, 405, 505};
          if (o.tile_cfg == 22) kind += 182;                                               // 232: the 1x1 GEMM kernel (gemm1x1.hip)
          else
          kind += base[(o.tile_cfg == 11 && o.cp.wino_split) ? 12 : o.tile_cfg] + ((o.tile_cfg >= 5 && o.has_x2) ? 2 : 0);
          if (o.tile_cfg == 11 && o.cp.wino_split == 2) kind += 20;                         // 575: the two-workgroups-per-CU split kernel (conv3x3_wino2.hip)
          if (o.tile_cfg >= 1 && o.tile_cfg <= 4 && o.cp.igemm_split) kind += 600;           // 651-654: the im2col tiles on their 3 x bf16 split instantiation
          WinoGeom wg;
          if (o.tile_cfg == 11 && wino_geometry(o.cp, &wg) && wg.NB != 1) kind += 10;      // 465: the four-image 8x8 tile
        }
        const ConvParams& c = o.cp;
        fl = 2.0 * c.B * c.Ho * c.Wo * (double)c.Cout * ((double)(c.C0 + c.C1) * c.ksize * c.ksize + (o.has_x2 ? c.x2_C0 + c.x2_C1 : 0));
        if (o.ksplit > 1) {      // split the op into its GEMM kernel and its split-K reduce kernel
          float a = 0.f;
          if (hipError_t e = hipEventElapsedTime(&a, ev[i], mid[i]); e != hipSuccess) { rc = hip_fail(e, "hipEventElapsedTime"); break; }
          red_ms = ms - a;
          ms = a;
        }
      } else if (o.kind == OP_ATTN) {
        fl = 4.0 * batch * (double)o.i0 * (double)o.i0 * o.i1;
      }
      if (w + 2 > max_ops) { set_error("op buffer too small"); rc = SR3_E_NOMEM; break; }
      op_ms[w] = ms; op_kind[w] = kind; op_flops[w] = fl; ++w;
      if (red_ms >= 0.f) { op_ms[w] = red_ms; op_kind[w] = 59; op_flops[w] = 0.0; ++w; }
    }
    *n_ops = w;
  }
  for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : mid) if (e) (void)hipEventDestroy(e);
  return rc;
}

int sr3_p_sample_step(float* x, const float* eps, const float* z, const float* ta, const float* tb, const float* tc1,
                      const float* tc2, const float* tsig, const int* step_dev, const int64_t* t_per_sample,
                      int step_host, int batch, int elems_per_image, void* stream) {
  if (!x || !eps || !ta || !tb || !tc1 || !tc2 || !tsig) { set_error("null argument"); return SR3_E_BADARG; }
  StepTables t{ta, tb, tc1, tc2, tsig};
  return p_sample_update(x, eps, z, t, step_dev, t_per_sample, step_host, batch, elems_per_image,
                         static_cast<hipStream_t>(stream));
}
int sr3_p_sample_step_ex(float* x, const float* eps, const float* z, const float* ta, const float* tb, const float* tc1,
                         const float* tc2, const float* tsig, const int* step_dev, const int64_t* t_per_sample,
                         int step_host, int batch, int elems_per_image, int clip_denoised, void* stream) {
  if (!x || !eps || !ta || !tb || !tc1 || !tc2 || !tsig) { set_error("null argument"); return SR3_E_BADARG; }
  StepTables t{ta, tb, tc1, tc2, tsig};
  return p_sample_update(x, eps, z, t, step_dev, t_per_sample, step_host, batch, elems_per_image,
                         static_cast<hipStream_t>(stream), clip_denoised != 0);
}
int sr3_step_decrement(int* step_dev, void* stream) { return step_decrement(step_dev, static_cast<hipStream_t>(stream)); }
int sr3_q_sample(const float* x0, const float* z, const float* ca, const float* cb, int batch, int elems_per_image,
                 float* out, void* stream) {
  if (!x0 || !z || !ca || !cb || !out) { set_error("null argument"); return SR3_E_BADARG; }
  return q_sample(x0, z, ca, cb, batch, elems_per_image, out, static_cast<hipStream_t>(stream));
}

int sr3_conv_f32(const float* src0, int C0, const float* src1, int C1, int B, int Hs, int Ws, int ups, int stride,
                 int ksize, int Cout, const float* w, const float* bias, const float* ss, int act, const float* film,
                 int film_stride, const float* res0, int RC0, const float* res1, int RC1, float* out, double* out_stats,
                 int tile_cfg, int ksplit, void* scratch, size_t scratch_bytes, void* stream) {
  if (!src0 || !w || !out) { set_error("null argument"); return SR3_E_BADARG; }
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.src0 = src0; c.src1 = src1; c.C0 = C0; c.C1 = src1 ? C1 : 0;
  c.B = B; c.Hs = Hs; c.Ws = Ws; c.ups = ups; c.stride = stride; c.ksize = ksize;
  const int pad = ksize / 2;
  c.Ho = ((Hs << ups) + 2 * pad - ksize) / stride + 1;
  c.Wo = ((Ws << ups) + 2 * pad - ksize) / stride + 1;
  c.Cout = Cout; c.w = w; c.bias = bias; c.ss = ss; c.act = act; c.film = film; c.film_stride = film_stride;
  c.res0 = res0; c.res1 = res1; c.RC0 = res0 ? RC0 : 0; c.RC1 = res1 ? RC1 : 0;
  c.out = out; c.ostat = out_stats; c.ksplit = 1;
  const bool wsplit = tile_cfg == 12 || tile_cfg == 13;     // tile 11 on the 3 x bf16 split instantiation (13: the 8 x 16 tile of conv3x3_wino2.hip)
  if (wsplit) { c.wino_split = tile_cfg - 11; tile_cfg = 11; }
  if (tile_cfg >= 14 && tile_cfg <= 17) { c.igemm_split = 1; tile_cfg -= 13; }     // the im2col tiles 1-4 on their 3 x bf16 split instantiation
  if (tile_cfg >= 18 && tile_cfg <= 22) {
    // ... with the weights pre-split into bf16 planes (what a plan does, in its derived buffer): derived here, behind the split-K
    // slabs in `scratch` (sr3_conv_scratch_bytes accounts for them).  22: the 1x1 GEMM kernel of gemm1x1.hip (64 x 128 tile)
    c.igemm_split = 1;
    if (tile_cfg == 22 && !gemm1x1_fits(c, 2)) { set_error("conv: the 1x1 GEMM kernel (tile 22) does not fit this problem"); return SR3_E_UNSUPPORTED; }
    if (tile_cfg <= 21) tile_cfg -= 17;
    const size_t slab = conv_splitk_bytes(c, tile_cfg, ksplit);
    const size_t wb = igemm_wsplit_floats(Cout, ksize * ksize, c.C0 + c.C1) * sizeof(float);
    if (!scratch || scratch_bytes < slab + wb) { set_error("conv: scratch too small for the pre-split weights (%zu < %zu)", scratch_bytes, slab + wb); return SR3_E_NOMEM; }
    float* q = reinterpret_cast<float*>(static_cast<char*>(scratch) + slab);
    const int rc = igemm_split_weights(w, Cout, ksize * ksize, c.C0 + c.C1, q, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    c.w_split = q;
    scratch_bytes = slab;
  }
  if (tile_cfg == 11 && (ksize != 3 || stride != 1)) { set_error("conv: the Winograd kernel does not fit this problem (3x3 stride 1 only)"); return SR3_E_UNSUPPORTED; }
  if (tile_cfg == 11) {
    // Winograd form through the per-op entry: the transformed filters are derived here, behind the split-K slabs in
    // `scratch` (sr3_conv_scratch_bytes accounts for them); a plan keeps them in its derived buffer instead.
    const size_t slab = conv_splitk_bytes(c, tile_cfg, ksplit);
    const size_t ub = wino_weight_floats(Cout, c.C0 + c.C1, wsplit) * sizeof(float);
    if (!scratch || scratch_bytes < slab + ub || ksize != 3) { set_error("conv: Winograd scratch too small (%zu < %zu)", scratch_bytes, slab + ub); return SR3_E_NOMEM; }
    float* u = reinterpret_cast<float*>(static_cast<char*>(scratch) + slab);
    const int rc = wino_transform_weights(w, Cout, c.C0 + c.C1, u, static_cast<hipStream_t>(stream), wsplit);
    if (rc) return rc;
    c.wino_u = u;
#ifdef SR3_WINO_ABLATIONS
    // tooling build only: room behind the filters for the kernel's phase time stamps (SR3_WINO_DBG=64, tools/wino_phases.py)
    if (scratch_bytes >= slab + ub + (1u << 20)) c.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(u) + ub);
#endif
    scratch_bytes = slab;
  }
  return conv_forward(c, tile_cfg, ksplit, static_cast<float*>(scratch), scratch_bytes, static_cast<hipStream_t>(stream));
}
int sr3_block_conv_f32(const float* src0, int C0, const float* src1, int C1, int B, int H, int W, int Cout,
                       const float* w, const float* bias, const float* ss, int act, const float* film, int film_stride,
                       const float* x2_src0, int x2_C0, const float* x2_src1, int x2_C1, const float* x2_w,
                       const float* x2_bias, float* out, double* out_stats, int tile_cfg, int ksplit, void* scratch,
                       size_t scratch_bytes, void* stream) {
  if (!src0 || !w || !out || !x2_src0 || !x2_w) { set_error("null argument"); return SR3_E_BADARG; }
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.src0 = src0; c.src1 = src1; c.C0 = C0; c.C1 = src1 ? C1 : 0;
  c.B = B; c.Hs = H; c.Ws = W; c.stride = 1; c.ksize = 3; c.Ho = H; c.Wo = W;
  c.Cout = Cout; c.w = w; c.bias = bias; c.ss = ss; c.act = act; c.film = film; c.film_stride = film_stride;
  c.out = out; c.ostat = out_stats; c.ksplit = 1;
  c.x2_src0 = x2_src0; c.x2_src1 = x2_src1; c.x2_C0 = x2_C0; c.x2_C1 = x2_src1 ? x2_C1 : 0; c.x2_w = x2_w; c.x2_bias = x2_bias;
  return conv_forward(c, tile_cfg, ksplit, static_cast<float*>(scratch), scratch_bytes, static_cast<hipStream_t>(stream));
}
int sr3_conv_dropout_f32(const float* src0, int C0, int B, int H, int W, int Cout, const float* w, const float* bias,
                         const float* ss, int act, const float* film, int film_stride, const float* res0, int RC0,
                         const float* x2_src0, int x2_C0, const float* x2_src1, int x2_C1, const float* x2_w,
                         const float* x2_bias, float* out, double* out_stats, int tile_cfg, int ksplit, void* scratch,
                         size_t scratch_bytes, unsigned drop_seed, float drop_p, void* stream) {
  if (!src0 || !w || !out || !ss) { set_error("null argument"); return SR3_E_BADARG; }
  if (drop_p < 0.f || drop_p >= 1.f) { set_error("drop_p out of range"); return SR3_E_BADARG; }
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.src0 = src0; c.C0 = C0;
  c.B = B; c.Hs = H; c.Ws = W; c.stride = 1; c.ksize = 3; c.Ho = H; c.Wo = W;
  c.Cout = Cout; c.w = w; c.bias = bias; c.ss = ss; c.act = act; c.film = film; c.film_stride = film_stride;
  c.res0 = res0; c.RC0 = res0 ? RC0 : 0;
  c.out = out; c.ostat = out_stats; c.ksplit = 1;
  if (x2_src0) {
    if (!x2_w) { set_error("x2_src0 needs x2_w"); return SR3_E_BADARG; }
    c.x2_src0 = x2_src0; c.x2_src1 = x2_src1; c.x2_C0 = x2_C0; c.x2_C1 = x2_src1 ? x2_C1 : 0; c.x2_w = x2_w; c.x2_bias = x2_bias;
  }
  // same mapping p -> (threshold, scale) as sr3_train_step
  c.drop_seed = drop_seed;
  dropout_consts(drop_p, &c.drop_thresh, &c.drop_scale);
  const bool wsplit = tile_cfg == 12;         // 12 = tile 11 on the 3 x bf16 split instantiation
  if (wsplit) { tile_cfg = 11; c.wino_split = 1; }
  if (tile_cfg == 11) {       // Winograd form: the transformed filters are derived here, behind the split-K slabs (as sr3_conv_f32)
    if (c.x2_w) { set_error("conv: the Winograd kernel has no fused 1x1 segment"); return SR3_E_UNSUPPORTED; }
    const size_t slab = conv_splitk_bytes(c, tile_cfg, ksplit);
    const size_t ub = wino_weight_floats(Cout, C0, wsplit) * sizeof(float);
    if (!scratch || scratch_bytes < slab + ub) { set_error("conv: Winograd scratch too small (%zu < %zu)", scratch_bytes, slab + ub); return SR3_E_NOMEM; }
    float* u = reinterpret_cast<float*>(static_cast<char*>(scratch) + slab);
    const int rc = wino_transform_weights(w, Cout, C0, u, static_cast<hipStream_t>(stream), wsplit);
    if (rc) return rc;
    c.wino_u = u;
    scratch_bytes = slab;
  }
  return conv_forward(c, tile_cfg, ksplit, static_cast<float*>(scratch), scratch_bytes, static_cast<hipStream_t>(stream));
}
unsigned sr3_dropout_threshold(float drop_p, float* scale_out) {
  unsigned t = 0;
  float s = 1.f;
  if (drop_p > 0.f && drop_p < 1.f) dropout_consts(drop_p, &t, &s);
  if (scale_out) *scale_out = s;
  return t;
}
size_t sr3_conv_scratch_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize, int tile_cfg, int ksplit) {
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.B = B; c.Ho = Ho; c.Wo = Wo; c.C0 = Cin; c.Cout = Cout; c.ksize = ksize;
  if (tile_cfg >= 11 && tile_cfg <= 13) {      // Winograd: the geometry (hence the split) needs the stride-1 input dims; + the derived filters
    c.Hs = Ho; c.Ws = Wo; c.stride = 1;
    return conv_splitk_bytes(c, 11, ksplit) + wino_weight_floats(Cout, Cin, tile_cfg >= 12) * sizeof(float);
  }
  size_t extra = 0;
  if (tile_cfg >= 14 && tile_cfg <= 17) { c.igemm_split = 1; tile_cfg -= 13; }
  if (tile_cfg == 22) {      // the GEMM kernel: 1x1 stride 1, or 3x3 stride 2 (ksize 3); pre-split weights behind the slabs
    c.igemm_split = 1;
    if (ksize == 3) { c.Hs = 2 * Ho; c.Ws = 2 * Wo; c.stride = 2; }
    else { c.Hs = Ho; c.Ws = Wo; c.stride = 1; }
    return conv_splitk_bytes(c, tile_cfg, ksplit) + igemm_wsplit_floats(Cout, ksize * ksize, Cin) * sizeof(float);
  }
  if (tile_cfg >= 18 && tile_cfg <= 21) {      // + the pre-split weights behind the slabs
    c.igemm_split = 1; tile_cfg -= 17;
    extra = igemm_wsplit_floats(Cout, ksize * ksize, Cin) * sizeof(float);
  }
  // the entry does not know the stride: take the larger of the stride-1 (halo kernel eligible) and the im2col sizing
  const size_t a = conv_splitk_bytes(c, tile_cfg, ksplit);
  c.Hs = Ho; c.Ws = Wo; c.stride = 1;
  const size_t b = conv_splitk_bytes(c, tile_cfg, ksplit);
  return (a > b ? a : b) + extra;
}
int sr3_groupnorm_stats_f32(const float* x, int B, int HW, int C, double* stat, void* stream) {
  if (!x || !stat) { set_error("null argument"); return SR3_E_BADARG; }
  return chan_stats(x, B, HW, C, stat, static_cast<hipStream_t>(stream));
}
int sr3_groupnorm_stats_slices(int B, int HW, int C) { return chan_stats_slices(B, HW, C); }
int sr3_conv_stats_slices(int B, int Hs, int Ws, int ups, int Cin, int Cout, int tile_cfg, int ksplit) {
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.B = B; c.Hs = Hs; c.Ws = Ws; c.ups = ups; c.stride = 1; c.ksize = 3; c.Ho = Hs << ups; c.Wo = Ws << ups;
  c.Cout = Cout; c.C0 = Cin;
  if (tile_cfg == 12 || tile_cfg == 13) { c.wino_split = tile_cfg - 11; tile_cfg = 11; }     // (13: the 8 x 16 tile, its own slice count)
  if (tile_cfg >= 14 && tile_cfg <= 17) { c.igemm_split = 1; tile_cfg -= 13; }
  if (tile_cfg >= 18 && tile_cfg <= 21) { c.igemm_split = 1; tile_cfg -= 17; }
  conv_pick(c, tile_cfg, ksplit);
  if (ksplit > 1) {
    const int rpb = splitk_rows_per_block(c, true);
    return rpb > 0 ? (c.Ho * c.Wo) / rpb : 0;
  }
  if (tile_cfg == 11) {
    WinoGeom wg;
    return wino_geometry(c, &wg) ? wino_stats_slices(wg) : 0;
  }
  HaloGeom g;
  if (tile_cfg < 5 || !halo_geometry(c, tile_cfg, &g)) return 0;
  return halo_stats_slices(g);
}
int sr3_groupnorm_fold_f32(const double* stat0, int C0, int T0, const double* stat1, int C1, int T1, int B, int HW,
                           int groups, const float* gamma, const float* beta, float eps, float* ss, void* stream) {
  if (!stat0 || !gamma || !beta || !ss) { set_error("null argument"); return SR3_E_BADARG; }
  return gn_finalize(stat0, C0, T0, stat1, stat1 ? C1 : 0, stat1 ? T1 : 0, B, HW, groups, gamma, beta, eps, ss,
                     static_cast<hipStream_t>(stream));
}
int sr3_attention_f32(const float* qkv, int B, int N, int C, float* out, void* stream) {
  if (!qkv || !out) { set_error("null argument"); return SR3_E_BADARG; }
  return attention_forward(qkv, B, N, C, out, static_cast<hipStream_t>(stream));
}
int sr3_attention_ex_f32(const float* qkv, int B, int N, int C, float* out, int split, void* stream) {
  if (!qkv || !out) { set_error("null argument"); return SR3_E_BADARG; }
  return attention_forward(qkv, B, N, C, out, static_cast<hipStream_t>(stream), split);
}
int sr3_film_embed_f32(int variant, int B, int inner, const float* level, const int64_t* timestep, const float* freq,
                       const float* w1, const float* b1, const float* w2, const float* b2, const float* wf,
                       const float* bf, int F, float* temb_scratch, float* film_out, void* stream) {
  EmbedParams e;
  memset(&e, 0, sizeof(e));
  e.variant = variant; e.B = B; e.inner = inner; e.level = level; e.tstep = timestep; e.freq = freq;
  e.w1 = w1; e.b1 = b1; e.w2 = w2; e.b2 = b2; e.wf = wf; e.bf = bf; e.F = F; e.temb = temb_scratch; e.film = film_out;
  return embed_forward(e, static_cast<hipStream_t>(stream));
}
int sr3_conv_in_f32(const float* a, int Ca, const float* b, int Cb, int B, int H, int W, const float* w,
                    const float* bias, int Cout, float* out, void* stream) {
  return conv_in_nchw(a, Ca, b, b ? Cb : 0, B, H, W, w, bias, Cout, out, nullptr, static_cast<hipStream_t>(stream));
}
int sr3_conv_out_f32(const float* x, const float* ss, int B, int H, int W, int C, const float* w, const float* bias,
                     int Cout, float* out, void* stream) {
  return conv_out_nchw(x, ss, B, H, W, C, w, bias, Cout, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
