// lds_dma.hpp — global -> LDS DMA (global_load_lds_*) for the staged filter kernels (atrous.hip, atrous_chain.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {
namespace {

// LDS-DMA issued from inline asm.  hipcc models `__builtin_amdgcn_global_load_lds` as an LDS store and
// puts `s_waitcnt vmcnt(0)` in front of every later ds_read that may alias it — with a ring buffer
// that is every read, which drains the prefetch each step.  Hidden in asm, the DMA is invisible to
// that pass and is ordered by hand: counted vmcnt + s_barrier before the reads (below).  M0 carries
// the wave-uniform LDS byte address; lane i lands at M0 + i*size.  One wait state is required between
// the SALU write of M0 and the LDS-DMA that reads it.
// `base` is a wave-uniform pointer (SGPR pair), `voff` the per-lane byte offset: no 64-bit VALU math.
// M0 is on the clobber list: the compiler must not assume a value it placed there (s_movrel indexing, sendmsg,
// its own LDS-DMA builtin) survives the statement.  clang warns that M0 is a reserved register; naming it is
// exactly the point, so that warning is silenced for these two functions only.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_b128(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

}  // namespace
}  // namespace rt
