// Host-side layout logic of the C ABI: the flat parameter arena (tensor table in
// module.parameters() order) and the carving of the caller's workspace for every GEMM mode.
// Pure C++ (no HIP): included by api.hip and, under AddressSanitizer + UBSan, by
// tools/host_sanitize.cpp in the CPU test job.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bilinear_hip.h"
#include "host_util.h"

namespace blh {

static constexpr int64_t ARENA_ALIGN = 64;   // floats (256 B)
static constexpr int64_t WS_ALIGN = 256;     // bytes

struct TensorInfo {
  char name[64];
  int64_t offset, rows, cols;
};

struct HeavyOffsets {
  int64_t w, b, gamma, beta;
  int fan_in;
};

struct ArenaLayout {
  std::vector<TensorInfo> tensors;
  std::vector<HeavyOffsets> heavy;
  int64_t dec_w, dec_b;
  int64_t total;
};

static int check_desc(const blh_model_desc* d) {
  if (!d) return BLH_ERR_INVALID_ARGUMENT;
  if (d->num_blocks < 0 || d->width <= 0 || d->in_features <= 0 || d->out_features <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (d->width % 64 != 0 || d->in_features % 4 != 0 || d->out_features % 4 != 0)
    return BLH_ERR_SHAPE;
  if (d->out_features > 64) return BLH_ERR_SHAPE;   // decode uses one 64-wide column tile
  if (1 + 2 * d->num_blocks > 32) return BLH_ERR_SHAPE;
  // (1 was round 1's mixed mode — fp32 tensors, operands rounded to bf16 on load — superseded by 4 and removed)
  if (d->gemm_dtype < 0 || d->gemm_dtype > 4 || d->gemm_dtype == 1) return BLH_ERR_INVALID_ARGUMENT;
  return BLH_OK;
}

static void heavy_prefix(int i, char* buf, size_t cap) {
  if (i == 0) snprintf(buf, cap, "encode");
  else snprintf(buf, cap, "bilinear.%d.%d", (i - 1) / 2, (i - 1) % 2);
}

static ArenaLayout make_layout(const blh_model_desc* d) {
  ArenaLayout L;
  int64_t off = 0;
  const int nh = 1 + 2 * d->num_blocks;
  auto add = [&](const char* prefix, const char* leaf, int64_t rows, int64_t cols) {
    TensorInfo t;
    snprintf(t.name, sizeof(t.name), "%s.%s", prefix, leaf);
    t.offset = off; t.rows = rows; t.cols = cols;
    L.tensors.push_back(t);
    const int64_t at = off;
    off = round_up(off + rows * cols, ARENA_ALIGN);
    return at;
  };
  for (int i = 0; i < nh; ++i) {
    char pre[48];
    heavy_prefix(i, pre, sizeof(pre));
    HeavyOffsets h;
    h.fan_in = (i == 0) ? d->in_features : d->width;
    h.w = add(pre, "0.weight", d->width, h.fan_in);
    h.b = add(pre, "0.bias", d->width, 1);
    h.gamma = add(pre, "1.weight", d->width, 1);
    h.beta = add(pre, "1.bias", d->width, 1);
    L.heavy.push_back(h);
  }
  L.dec_w = add("decode", "weight", d->out_features, d->width);
  L.dec_b = add("decode", "bias", d->out_features, 1);
  L.total = off;
  return L;
}

// ------------------------------------------------------------ workspace ----
struct Workspace {
  std::vector<float*> Z, A;       // per heavy: pre-BN output, activation (skip added)
  std::vector<float*> bn_saved;   // per heavy: [4][W] mean, invstd, scale, shift
  std::vector<uint32_t*> keep;    // per heavy: dropout keep bits [ceil(B/8)][W/4] words (bn_f32.hip)
  float* stat_part;               // [tiles_m][2][W]
  float* G0; float* G1;
  std::vector<float*> dZ;         // per heavy stage: no buffer is reused inside one backward, so
                                  // the side-stream weight gradients impose no wait on the main stream
  float* bn_part;                 // [chunks][2][W]
  float* dz_colsum_part;          // [stage][chunks][W]
  float* slabs;                   // split-K partial products
  float* dpred;                   // [B][out]
  float* loss_part;               // [4096]
  double* sumsq_part;             // [1024]
  double* sumsq_fold;             // [SUMSQ_FOLD_PARTS]: partials written by the gradient producers (fused step)
  float* colsum_part;             // [ceil(B/256)][out]
  double* sync_buf;               // [2][W] fp64 (SyncBN exchange; also used as float [2][W])
  float* enc_slabs;                // split-K slabs of the encode weight gradient (main stream beside the side stream)
  float* dec_bias_part;           // [blocks][out] partial sums of dpred (fused step)
  // gemm_dtype 3: max |value| partials of the GEMM operand tensors (see gemm_f16x2_kernel.h)
  std::vector<float*> amax_A;     // per heavy stage: activation A_l (written by bn_apply)
  float* amax_dZ[2];              // dZ / dZ2 (written by bn_bwd_apply)
  float* amax_W;                  // [nh][WAMAX_PARTS], hidden weights (stage 0 unused)
  int amax_parts;
  int64_t bytes;
};

struct Splits { int splits, k_per; };
// every reduction slab is a whole number of K tiles of the widest kernel (gemm_f32_ring.h: 64)
static constexpr int64_t SPLIT_GRAIN = 64;
static Splits pick_splits(int64_t batch, int64_t tiles) {
  int64_t want = std::max<int64_t>(1, ceil_div(256, tiles));
  int64_t max_splits = std::max<int64_t>(1, batch / 128);
  int64_t s = std::min(want, max_splits);
  int64_t k_per = round_up(ceil_div(batch, s), SPLIT_GRAIN);
  s = ceil_div(batch, k_per);
  return Splits{(int)s, (int)k_per};
}

// Small-batch Linear forward / dgrad (M = batch rows, N = W columns, reduction K): when the
// 128x128 output tiles cover less than half of the 256 CUs, split the reduction so that about
// one workgroup per CU is in flight.  Returns splits == 1 for the ordinary path.
// Half-chip batches (more than 128 and at most 256 tiles of 64 x 128: 1024 < batch <= 2048 at W = 1024) run the hidden
// Linears and their data gradients on 64-row tiles, one per CU, instead of 128-row tiles on half of the CUs (the
// forward GEMM took the same 66 us at 2048 rows as at 4096) or a split reduction with its finishing kernels: batch
// 2048 0.797 -> 0.67 ms per step.  Exact fp32 only (the 64 x 128 kernel has no split-precision form).
static bool mid_tile64(int64_t batch, int W, int dtype) {
  const int64_t tiles64 = ceil_div(batch, 64) * ceil_div(W, 128);
  // (more tiles than CUs lose: 272 tiles at 2176 rows 0.87 ms; 384 / 512 tiles at 3072 / 4096 rows 0.964 / 1.042 ms
  //  against 0.955 / 1.029 on 128-row tiles)
  return dtype == 0 && tiles64 > 128 && tiles64 <= 256;
}
static Splits small_m_splits(int64_t batch, int W, int K, int dtype = -1) {
  const int64_t tiles = ceil_div(batch, 128) * ceil_div(W, 128);
  if (tiles >= 128 || K < 64 || mid_tile64(batch, W, dtype)) return Splits{1, K};
  int64_t s = std::min<int64_t>(ceil_div(256, tiles), K / SPLIT_GRAIN);
  int64_t k_per = round_up(ceil_div(K, s), SPLIT_GRAIN);
  s = ceil_div(K, k_per);
  return Splits{(int)s, (int)k_per};
}

// decode forward: split W so that (B/128) * splits is about one workgroup per CU
static Splits decode_fwd_splits(int64_t batch, int W) {
  int64_t want = std::max<int64_t>(1, ceil_div(256, ceil_div(batch, 128)));
  int64_t s = std::min<int64_t>(want, std::max<int64_t>(1, W / 128));
  int64_t k_per = round_up(ceil_div(W, s), SPLIT_GRAIN);
  s = ceil_div(W, k_per);
  return Splits{(int)s, (int)k_per};
}

static int64_t slab_floats(const blh_model_desc* d, int64_t batch) {
  const int64_t W = d->width;
  const Splits hs = pick_splits(batch, ceil_div(W, 128) * ceil_div(W, 128));
  const Splits es = pick_splits(batch, ceil_div(W, 128) * ceil_div(d->in_features, 32));
  const Splits ds = pick_splits(batch, ceil_div(d->out_features, 64) * ceil_div(W, 128));
  int64_t m = hs.splits * W * W;
  m = std::max(m, es.splits * W * (int64_t)d->in_features);
  m = std::max(m, ds.splits * (int64_t)d->out_features * W);
  m = std::max(m, decode_fwd_splits(batch, d->width).splits * batch * d->out_features);
  m = std::max(m, small_m_splits(batch, d->width, d->width).splits * batch * W);
  return m;
}

static Workspace carve(const blh_model_desc* d, int64_t batch, void* base) {
  Workspace ws;
  const int nh = 1 + 2 * d->num_blocks;
  const int64_t W = d->width;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += round_up(bytes, WS_ALIGN);
    return r;
  };
  const int64_t act = batch * W * (int64_t)sizeof(float);
  for (int i = 0; i < nh; ++i) ws.Z.push_back((float*)take(act));
  for (int i = 0; i < nh; ++i) ws.A.push_back((float*)take(act));
  for (int i = 0; i < nh; ++i) ws.bn_saved.push_back((float*)take(4 * W * sizeof(float)));
  for (int i = 0; i < nh; ++i) ws.keep.push_back((uint32_t*)take(ceil_div(batch, 8) * (W / 4) * 4));
  ws.stat_part = (float*)take(ceil_div(batch, 64) * 2 * W * sizeof(float));
  ws.G0 = (float*)take(act);
  ws.G1 = (float*)take(act);
  for (int i = 0; i < nh; ++i) ws.dZ.push_back((float*)take(act));
  const int64_t chunks = ew_num_row_chunks(batch);
  ws.bn_part = (float*)take(chunks * 2 * W * sizeof(float));
  ws.dz_colsum_part = (float*)take((int64_t)nh * chunks * W * sizeof(float));
  ws.slabs = (float*)take(slab_floats(d, batch) * sizeof(float));
  ws.dpred = (float*)take(batch * d->out_features * sizeof(float));
  ws.loss_part = (float*)take(4096 * sizeof(float));
  ws.sumsq_part = (double*)take(SUMSQ_MAX_PARTS * sizeof(double));
  ws.sumsq_fold = (double*)take(SUMSQ_FOLD_PARTS * sizeof(double));
  ws.colsum_part = (float*)take(ceil_div(batch, 256) * d->out_features * sizeof(float));
  ws.sync_buf = (double*)take(2 * W * sizeof(double));
  {   // the encode weight gradient runs on the main stream while the shared slab buffer may still be in use by
      // wgrad(1) on the side stream: it takes slabs of its own
    const int64_t M = W, N = d->in_features;
    const Splits sp = pick_splits(batch, ceil_div(M, 128) * ceil_div(N, 32));
    ws.enc_slabs = sp.splits > 1 ? (float*)take(sp.splits * M * N * sizeof(float)) : nullptr;
  }
  ws.dec_bias_part = (float*)take(1026 * d->out_features * sizeof(float));
  ws.amax_parts = 0;
  ws.amax_dZ[0] = ws.amax_dZ[1] = ws.amax_W = nullptr;
  if (d->gemm_dtype == 3) {
    ws.amax_parts = ew_num_amax_parts(batch, (int)W);
    for (int i = 0; i < nh; ++i) ws.amax_A.push_back((float*)take(ws.amax_parts * sizeof(float)));
    ws.amax_dZ[0] = (float*)take(ws.amax_parts * sizeof(float));
    ws.amax_dZ[1] = (float*)take(ws.amax_parts * sizeof(float));
    ws.amax_W = (float*)take((int64_t)nh * WAMAX_PARTS * sizeof(float));
  } else {
    for (int i = 0; i < nh; ++i) ws.amax_A.push_back(nullptr);
  }
  ws.bytes = off;
  return ws;
}

// small scratch for entry points that take only a workspace pointer (no model)
struct Scratch {
  float* loss_part;
  double* sumsq_part;
};
static constexpr int64_t SCRATCH_BYTES = 4096 * sizeof(float) + SUMSQ_MAX_PARTS * sizeof(double);
static Scratch carve_scratch(void* base) {
  Scratch s;
  s.loss_part = (float*)base;
  s.sumsq_part = (double*)((char*)base + 4096 * sizeof(float));
  return s;
}


// ---- gemm_dtype 4 ("bf16s"): every [B,W] tensor, the input and a parameter shadow in bf16 ----
struct WorkspaceH {
  uint16_t* wsh;                    // bf16 image of the whole parameter arena (refreshed per forward)
  uint16_t* xh;                     // [B][in] network input
  std::vector<uint16_t*> Z, A, dZ;  // per heavy stage, [B][W]
  std::vector<float*> bn_saved;     // per heavy stage [4][W]
  std::vector<uint32_t*> keep;      // per heavy stage: dropout keep bits [ceil(B/4)][W/8] (bn_bf16.hip)
  uint16_t* G0; uint16_t* G1;
  float* stat_part; float* bn_part; float* dz_colsum_part; float* slabs;
  int64_t slab_cap;                 // floats in `slabs` (every split launch is checked against it)
  float* bslabs;                    // [hidden stages][slabs][W][W]: the batched weight gradient (null: not planned)
  float* dpred; uint16_t* dpredh;   // [B][out] fp32 and its bf16 image
  float* loss_part; double* sumsq_part; float* colsum_part;
  double* sync_buf;                 // [2][W] fp64 (SyncBN exchange; its first 2W floats in backward)
  float* dec_bias_part;             // [blocks][out] partial sums of dpred (fused decode + MSE kernel)
  uint16_t* wdT;                    // [W][64]: the decode weight's image for the one-pass decode's second phase (decode_wdT_dev.h)
  int64_t bytes;
};

// Batch slabs of a bf16-storage weight gradient dW[M][N] = dZ^T act (reduction over the batch).
// Outputs with at least 64 tiles of 256 x 256 (W >= 2048) run on the 256 x 256 kernel
// (gemm_bf16s_256.h: 1.43 PFLOP/s against 0.7-0.9 for the 128 x 128 kernel at W = 2048) with at most
// 4 equal slabs of whole 128-deep steps; smaller outputs would need 16 slabs to fill the chip with
// such tiles (64 MB of fp32 slabs at W = 1024) and stay on the 128 x 128 kernel with about one
// workgroup per CU, slabs of whole 128-deep K tiles.
static int64_t wgrad256_min_tiles() {
  static const int64_t v = [] {            // BLH_WGRAD256_MIN_TILES: developer tuning knob
    const char* e = std::getenv("BLH_WGRAD256_MIN_TILES");
    const int n = e ? std::atoi(e) : 0;
    return (int64_t)((n >= 1 && n <= 4096) ? n : 64);
  }();
  return v;
}
static Splits wgrad_plan_h(int64_t M, int64_t N, int64_t batch) {
  if (M % 256 == 0 && N % 256 == 0 && (M / 256) * (N / 256) >= wgrad256_min_tiles()) {
    int64_t s = std::max<int64_t>(1, 256 / ((M / 256) * (N / 256)));
    while (s > 1 && batch % (s * 128) != 0) s >>= 1;
    if (batch % (s * 128) == 0) return Splits{(int)s, (int)(batch / s)};
  }
  Splits sp = pick_splits(batch, ceil_div(M, 128) * ceil_div(N, 128));
  sp.k_per = (int)round_up(sp.k_per, 128);
  sp.splits = (int)ceil_div(batch, sp.k_per);
  return sp;
}

// Batched weight gradient (r03): the weight gradients of all `items` hidden W x W stages in ONE launch of
// the 256 x 256 kernel (gemm_bf16s_256.h, grid z = stage x slab) after the backward loop, with as few
// batch slabs per stage as make about 256 workgroups: 8 stages x 2 slabs at configs[2] (each workgroup
// 8192 rows deep: 25 us per stage against 42 alone / 60-100 in the step for the per-stage 128 x 128 plan),
// 16 stages x 1 slab at configs[4] (no slabs at all, and no 256 x 256 weight gradient beside the
// BatchNorm chain, which cannot run beside one: 7.73 -> 7.40 ms).  Under a bucket hook `items` is a group of
// WGRAD_HOOK_GROUP stages (api.hip: backward_h).  {0, 0}: not applicable (the per-stage plan above is used).
static constexpr int WGRAD_HOOK_GROUP = 4;     // stages per batched launch under a bucket hook
static Splits wgrad_batched_plan_h(int64_t W, int64_t batch, int items, bool hook = false) {
  static const bool off = std::getenv("BLH_NO_BATCHED_WGRAD") != nullptr;     // (developer knob, read once)
  if (items < 2 || W % 256 != 0 || batch % 128 != 0 || off) return Splits{0, 0};
  const int64_t tiles = (W / 256) * (W / 256);
  int64_t s = 1;
  while (s < 8 && tiles * items * (s * 2) <= 256) s *= 2;
  while (s > 1 && batch % (s * 128) != 0) s >>= 1;
  // (round 4 asked for workgroups at least 4096 rows deep and >= 224 of them; round 5's sweep of 2 x 1024 and 4 x 1024
  //  at 1024 .. 6144 rows — tools_dev/bf16_small_batch_ab.sh, profiles/r05_bf16_small_batch.md — has ONE launch beat the
  //  per-stage launches + slab sums down to 512 rows per workgroup (every winning case had 256 workgroups): 4 x 1024 at 1024 / 2048 / 3072 /
  //  4096 rows 0.659 / 0.692 / 0.720 / 0.736 -> 0.611 / 0.644 / 0.673 / 0.719 ms; 2 x 1024 at 4096 rows +1 %)
  static const int64_t min_rows = [] {      // (developer knob, read once: rows per workgroup below which the plan is refused)
    const char* e = std::getenv("BLH_WGRAD_BATCHED_MIN_ROWS");
    return e ? (int64_t)std::atoll(e) : (int64_t)512;
  }();
  static const int64_t min_wgs = [] {       // (developer knob, read once: workgroups below which the plan is refused)
    const char* e = std::getenv("BLH_WGRAD_BATCHED_MIN_WGS");
    return e ? (int64_t)std::atoll(e) : (int64_t)224;     // (what gemm_bf16s_pick_tile asks of a 256 x 256 launch)
  }();
  // (under a bucket hook the round-4 depth stays: groups become ready together, and a 2-block model whose four hidden
  //  stages form ONE group would put the whole arena on the wire after the last stage — no overlap left)
  if (tiles * items * s < min_wgs || batch / s < (hook ? std::max<int64_t>(min_rows, 4096) : min_rows)) return Splits{0, 0};
  return Splits{(int)s, (int)(batch / s)};
}

// (for the batch AND for its multiple-of-8 part, which is what a ragged batch launches — step_bf16s.hip: wgrad_h.  The two
//  plans differ: 385 rows round their slab depth up to 256 and need 2 slabs, the 384 rows actually launched keep 128 and
//  need 3 — sized for the first, the slabs of every ragged batch above 384 rows ran 262 KB (W = 256) .. 8 MB (W = 1024)
//  past this buffer, over the loss / decode-bias partials behind it and, at the end of the workspace, out of it: found by
//  tests/shape_fuzz.py in round 6, the fused step returned a zero loss and a wrong decode-bias gradient at such batches)
static int64_t slab_floats_h(const blh_model_desc* d, int64_t batch) {
  const int64_t W = d->width;
  int64_t m = 0;
  for (const int64_t b : {batch, batch & ~(int64_t)7}) {
    if (b <= 0) continue;
    m = std::max(m, wgrad_plan_h(W, W, b).splits * W * W);
    m = std::max(m, wgrad_plan_h(W, d->in_features, b).splits * W * (int64_t)d->in_features);
    m = std::max(m, wgrad_plan_h(d->out_features, W, b).splits * (int64_t)d->out_features * W);
  }
  return m;
}

static WorkspaceH carve_h(const blh_model_desc* d, int64_t batch, void* base) {
  WorkspaceH ws;
  const int nh = 1 + 2 * d->num_blocks;
  const int64_t W = d->width;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += round_up(bytes, WS_ALIGN);
    return r;
  };
  const int64_t act = batch * W * 2;
  ws.wsh = (uint16_t*)take(make_layout(d).total * 2);
  // (the two parameter images FIRST, at offsets that do not depend on the batch: a caller may run batches of different
  //  sizes in one workspace, and an image the previous step's Adam kernel left — BLH_OPT_PERSISTENT_SHADOW — must be
  //  where the next step looks for it.  Round 6 found wdT behind the batch-sized buffers: after a change of the batch
  //  size the one-pass decode read its weight from wherever the OTHER batch size had put it — a stale or never-written
  //  image, silently wrong decode data gradients: 3e-2 of the parameters after seven steps, tools_dev/shadow_batch_change.py)
  ws.wdT = (uint16_t*)take(W * 64 * 2);
  ws.xh = (uint16_t*)take(batch * d->in_features * 2);
  for (int i = 0; i < nh; ++i) ws.Z.push_back((uint16_t*)take(act));
  for (int i = 0; i < nh; ++i) ws.A.push_back((uint16_t*)take(act));
  for (int i = 0; i < nh; ++i) ws.dZ.push_back((uint16_t*)take(act));
  for (int i = 0; i < nh; ++i) ws.bn_saved.push_back((float*)take(4 * W * sizeof(float)));
  for (int i = 0; i < nh; ++i) ws.keep.push_back((uint32_t*)take(ceil_div(batch, 4) * (W / 8) * 4));
  ws.G0 = (uint16_t*)take(act);
  ws.G1 = (uint16_t*)take(act);
  ws.stat_part = (float*)take(ceil_div(batch, 64) * 2 * W * sizeof(float));
  const int64_t chunks = ew_num_row_chunks_h(batch);
  ws.bn_part = (float*)take(chunks * 2 * W * sizeof(float));
  ws.dz_colsum_part = (float*)take((int64_t)nh * chunks * W * sizeof(float));
  ws.slab_cap = slab_floats_h(d, batch);
  ws.slabs = (float*)take(ws.slab_cap * sizeof(float));
  {   // slabs of the batched weight gradient: all hidden stages (no hook) or a group of up to four (hook)
    const int hidden = 2 * d->num_blocks;
    int64_t need = 0;
    for (int items : {hidden, std::min(hidden, WGRAD_HOOK_GROUP), hidden % WGRAD_HOOK_GROUP}) {
      for (bool hook : {false, true}) {
        const Splits bp = wgrad_batched_plan_h(W, batch, items, hook);
        if (bp.splits > 1) need = std::max(need, (int64_t)items * bp.splits * W * W);
      }
    }
    ws.bslabs = need ? (float*)take(need * sizeof(float)) : nullptr;
  }
  ws.dpred = (float*)take(batch * d->out_features * sizeof(float));
  ws.dpredh = (uint16_t*)take(batch * d->out_features * 2);
  ws.loss_part = (float*)take(4096 * sizeof(float));
  // (SUMSQ_FOLD_PARTS_H slots: the fused step's batched slab sum leaves up to that many norm partials here, r06)
  ws.sumsq_part = (double*)take(SUMSQ_FOLD_PARTS_H * sizeof(double));
  ws.colsum_part = (float*)take(ceil_div(batch, 256) * d->out_features * sizeof(float));
  ws.sync_buf = (double*)take(2 * W * sizeof(double));
  ws.dec_bias_part = (float*)take(1026 * d->out_features * sizeof(float));
  ws.bytes = off;
  return ws;
}


}  // namespace blh
