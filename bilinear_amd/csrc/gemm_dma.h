// Pieces every LDS-DMA GEMM kernel of this library shares: the 16-byte zero page, the inline-asm
// LDS-DMA issue (per-lane pointer form; the scalar-base form lives in gemm_f32_ring.h) and the
// XCD-aware workgroup -> tile maps.
#pragma once
#include "common.h"

namespace blh {

// ---- global -> LDS by LDS-DMA (global_load_lds_dwordx4), PIPE == 3 ---------------------
// The DMA writes LDS linearly (wave-uniform base + lane*16 B), so tiles are unpadded and the
// bank-conflict fix for the ROWK fragment reads is an XOR swizzle applied to the per-lane
// SOURCE address (chunk c of row r lands in slot c ^ (r & 7)) and again on the read.
// Out-of-range chunks read a 16-byte zero page instead of being masked (a masked lane would
// leave stale LDS bytes).
static __device__ float4 g_zero16;

// One LDS-DMA of 16 B per lane issued from inline asm, so that hipcc does not know a DMA is in
// flight: with the builtin it inserts a conservative `s_waitcnt vmcnt(0)` in front of every
// ds_read that might alias the DMA destination, which drains the ring each K tile.  M0 (LDS
// base of the wave's 1 KiB piece) is written in the same statement that uses it; nothing else
// in these kernels reads M0, so it is declared clobbered rather than saved and restored.
__device__ __forceinline__ void lds_dma16_asm(const float* gsrc, uint32_t lds_byte_addr_uniform) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_byte_addr_uniform)
      : "memory", "m0");
}

// XCD-aware, bijective remap of the linear workgroup id: workgroups b, b+8, ...
// share an XCD (and its 4 MiB L2); give each XCD a contiguous range of tiles so
// neighbouring tiles (same A row panel) hit the same L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Split grids (the weight gradient: gridDim.z = S batch slabs of the same tiles_m x tiles_n output).
// With xcd_remap alone every XCD owns one band of output rows for ALL slabs, i.e. it reads its
// column panel of the first operand and the WHOLE second operand: 168 MB of HBM-side reads per
// 1024^2 launch at B = 4096 against 50 MB algorithmic (profiles/r02_pmc_gemm.md).  Here an XCD owns
// ONE slab (S | 8: XCD j -> slab j % S) and, when S < 8, one of the 8 / S bands of output rows:
// each XCD then reads a 1/S slice of the batch rows once.  Returns false (and leaves tile / slab
// alone) when the grid does not divide that way.  Assumes what xcd_remap assumes: workgroups are
// handed to the XCDs round-robin in linear order (gridDim.x % 8 == 0 keeps that independent of z).
__device__ inline bool xcd_remap_split(int bx, int bz, int gx, int S, int tiles_m, int tiles_n,
                                       int* tile, int* slab) {
  if ((gx & 7) || S < 2 || S > 8 || (8 % S) || (tiles_m % (8 / S)) || tiles_m * tiles_n != gx) return false;
  const int xcd = bx & 7, idx = bx >> 3;           // idx < gx / 8
  const int slot = bz * (gx >> 3) + idx;           // < gx * S / 8 workgroups of this XCD
  const int band_rows = tiles_m / (8 / S);         // output-tile rows per band
  *slab = xcd % S;
  *tile = ((xcd / S) * band_rows + slot / tiles_n) * tiles_n + slot % tiles_n;
  return true;
}

}  // namespace blh
