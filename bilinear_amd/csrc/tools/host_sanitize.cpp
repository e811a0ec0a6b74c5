// Host-side sanitizer driver (CPU test job): compiles the layout / workspace-carving logic of the
// C ABI (api_layout.h: pure C++) under AddressSanitizer + UBSan and walks it — arena layout and
// tensor table, workspace carving for every GEMM mode over many shapes with the carved pointers
// checked to be aligned, disjoint, in order and inside the reported size — so that an overflow or
// an out-of-bounds step in that arithmetic shows up as a sanitizer report or a failed check.
// Build + run: make -C bilinear_amd/csrc sanitize   (g++, no HIP, no GPU).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../api_layout.h"

using namespace blh;

#define CHECK(cond)                                                                       \
  do {                                                                                    \
    if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; }    \
  } while (0)

struct Span { const char* p; int64_t bytes; };

static bool spans_ok(std::vector<Span>& v, const char* base, int64_t total) {
  const char* prev_end = base;
  for (const Span& s : v) {
    if (!s.p) continue;
    if (((uintptr_t)s.p - (uintptr_t)base) % WS_ALIGN != 0) return false;
    if (s.p < prev_end || s.p + s.bytes > base + total) return false;
    prev_end = s.p + s.bytes;
  }
  return true;
}

// bf16 storage: a ragged batch launches its weight-gradient GEMMs over batch & ~7 rows (step_bf16s.hip: wgrad_h); the slab
// buffer must hold THAT plan's slabs as well as the full batch's (round 6: it held only the latter — 385 rows need 2
// slabs, the 384 launched 3 — and every ragged batch above 384 rows wrote past it)
static int check_ragged_slab_plans() {
  for (int w : {128, 256, 512, 1024, 2048})
    for (int64_t b = 2; b <= 20000; b += (b < 4200 ? 1 : 97)) {
      blh_model_desc d{2, w, 32, 48, 4};
      const int64_t cap = slab_floats_h(&d, b);
      for (const int64_t rows : {b, b & ~(int64_t)7}) {
        if (rows <= 0) continue;
        CHECK(wgrad_plan_h(w, w, rows).splits * (int64_t)w * w <= cap);
        CHECK(wgrad_plan_h(w, 32, rows).splits * (int64_t)w * 32 <= cap);
        CHECK(wgrad_plan_h(48, w, rows).splits * (int64_t)48 * w <= cap);
      }
    }
  return 0;
}

int main() {
  if (check_ragged_slab_plans()) return 1;
  const int widths[] = {64, 128, 256, 1024, 2048, 4096};
  const int64_t batches[] = {2, 30, 64, 257, 4096, 4100, 16384, 131072};
  char* const base = reinterpret_cast<char*>(uintptr_t(1) << 40);   // never dereferenced
  for (int nb = 0; nb <= 15; nb += 3)
    for (int w : widths)
      for (int mode = 0; mode <= 4; ++mode) {
        blh_model_desc d{nb, w, 32, 48, mode};
        if (mode == 1) {   // round 1's mixed mode: removed, refused
          CHECK(check_desc(&d) == BLH_ERR_INVALID_ARGUMENT);
          continue;
        }
        CHECK(check_desc(&d) == BLH_OK);
        const ArenaLayout L = make_layout(&d);
        const int nh = 1 + 2 * nb;
        CHECK((int)L.tensors.size() == 4 * nh + 2 && (int)L.heavy.size() == nh);
        int64_t prev_end = 0;
        for (const TensorInfo& t : L.tensors) {
          CHECK(std::strlen(t.name) > 0 && std::strlen(t.name) < sizeof(t.name));
          CHECK(t.offset >= prev_end && t.offset % ARENA_ALIGN == 0);
          prev_end = t.offset + t.rows * t.cols;
        }
        CHECK(prev_end <= L.total && L.total % ARENA_ALIGN == 0);
        for (int64_t b : batches) {
          if (mode == 4) {
            if (w % 128 != 0) continue;
            const WorkspaceH ws = carve_h(&d, b, base);
            CHECK(ws.bytes == carve_h(&d, b, nullptr).bytes && ws.bytes > 0);
            std::vector<Span> v;
            v.push_back({(char*)ws.wsh, L.total * 2});
            v.push_back({(char*)ws.wdT, (int64_t)w * 64 * 2});
            // the parameter images sit where every batch size finds them (an image kept across steps must survive a
            // change of the batch size on one workspace)
            CHECK((char*)ws.wsh == base && ws.wdT == carve_h(&d, 2, base).wdT && ws.wdT == carve_h(&d, 65536, base).wdT);
            v.push_back({(char*)ws.xh, b * 32 * 2});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.Z[i], b * w * 2});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.A[i], b * w * 2});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.dZ[i], b * w * 2});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.bn_saved[i], 4 * w * 4});
            v.push_back({(char*)ws.G0, b * w * 2});
            v.push_back({(char*)ws.G1, b * w * 2});
            v.push_back({(char*)ws.stat_part, ceil_div(b, 64) * 2 * w * 4});
            v.push_back({(char*)ws.bn_part, (int64_t)ew_num_row_chunks_h(b) * 2 * w * 4});
            v.push_back({(char*)ws.dz_colsum_part, (int64_t)nh * ew_num_row_chunks_h(b) * w * 4});
            v.push_back({(char*)ws.slabs, slab_floats_h(&d, b) * 4});
            CHECK(ws.slab_cap == slab_floats_h(&d, b));
            v.push_back({(char*)ws.dpred, b * 48 * 4});
            v.push_back({(char*)ws.dpredh, b * 48 * 2});
            CHECK(spans_ok(v, base, ws.bytes));
          } else {
            const Workspace ws = carve(&d, b, base);
            CHECK(ws.bytes == carve(&d, b, nullptr).bytes && ws.bytes > 0);
            std::vector<Span> v;
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.Z[i], b * w * 4});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.A[i], b * w * 4});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.bn_saved[i], 4 * w * 4});
            v.push_back({(char*)ws.stat_part, ceil_div(b, 64) * 2 * w * 4});
            v.push_back({(char*)ws.G0, b * w * 4});
            v.push_back({(char*)ws.G1, b * w * 4});
            for (int i = 0; i < nh; ++i) v.push_back({(char*)ws.dZ[i], b * w * 4});
            v.push_back({(char*)ws.bn_part, (int64_t)ew_num_row_chunks(b) * 2 * w * 4});
            v.push_back({(char*)ws.dz_colsum_part, (int64_t)nh * ew_num_row_chunks(b) * w * 4});
            v.push_back({(char*)ws.slabs, slab_floats(&d, b) * 4});
            v.push_back({(char*)ws.dpred, b * 48 * 4});
            CHECK(spans_ok(v, base, ws.bytes));
            // every split plan covers its reduction range with whole K tiles
            const Splits hs = pick_splits(b, ceil_div(w, 128) * ceil_div(w, 128));
            CHECK(hs.splits >= 1 && (int64_t)hs.k_per * hs.splits >= b && hs.k_per % SPLIT_GRAIN == 0);
            CHECK((int64_t)hs.k_per * (hs.splits - 1) < b);
            const Splits fs = small_m_splits(b, w, w);
            CHECK(fs.splits >= 1 && (int64_t)fs.k_per * fs.splits >= w);
          }
        }
      }
  blh_model_desc bad{2, 1000, 32, 48, 0};
  CHECK(check_desc(&bad) == BLH_ERR_SHAPE);
  blh_model_desc bad2{2, 1024, 32, 48, 9};
  CHECK(check_desc(&bad2) == BLH_ERR_INVALID_ARGUMENT);
  CHECK(check_desc(nullptr) == BLH_ERR_INVALID_ARGUMENT);
  std::printf("host sanitize ok\n");
  return 0;
}
