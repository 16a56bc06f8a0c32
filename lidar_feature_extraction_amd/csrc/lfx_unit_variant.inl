// lfx_unit_variant.inl -- the unit kernels of ONE variant of the parameters (LFX_VARIANT, see UnitVariant in
// lfx_kernels_unit.hpp) and their launchers; included by lfx_unit_v0.hip ... lfx_unit_v3.hip, which are four translation
// units so that they build side by side and an edit that concerns one variant recompiles one of them.
#include "lfx_internal.hpp"
#include "lfx_kernels_unit.hpp"

#ifndef LFX_VARIANT
#error "LFX_VARIANT = 0 .. 3"
#endif

namespace lfx_host
{

static_assert(lfx::UnitVariant<LFX_VARIANT>::kPT == kUnitVariantPadding[LFX_VARIANT], "the host sizes the record slots from this table (lfx_create)");

#define LFX_CAT2(a, b) a##b
#define LFX_CAT(a, b) LFX_CAT2(a, b)

// chunks: 3, 4, 5, 6 or lfx::kUnitMaxChunks (the span variant the context picked for its longest ring)
// holes: the form for grids whose invalid returns are (0, 0, 0) records (a.xform = grid_count_kernel's table then)
void LFX_CAT(launch_unit_org_v, LFX_VARIANT)(int chunks, bool xf, bool holes, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitOrgArgs & a)
{
  constexpr int V = LFX_VARIANT;
  void (*kern)(lfx::Params, uint32_t, uint32_t, uint32_t, uint32_t, const uint8_t *, const uint32_t *, uint32_t *,
    const lfx::UnitTables *, const uint32_t *, const uint32_t *) = nullptr;
#define LFX_PICK_ORG(XFV) \
  (chunks == 5 ? &lfx::ring_unit_org_kernel<V, 5, XFV> : chunks == 4 ? &lfx::ring_unit_org_kernel<V, 4, XFV> : \
   chunks == 3 ? &lfx::ring_unit_org_kernel<V, 3, XFV> : chunks == 6 ? &lfx::ring_unit_org_kernel<V, 6, XFV> : \
   &lfx::ring_unit_org_kernel<V, lfx::kUnitMaxChunks, XFV>)
  kern = xf ? LFX_PICK_ORG(true) : LFX_PICK_ORG(false);
#undef LFX_PICK_ORG
  if (holes) {
    kern = chunks == 5 ? &lfx::ring_unit_org_kernel<V, 5, false, true> : chunks == 4 ? &lfx::ring_unit_org_kernel<V, 4, false, true> :
      chunks == 3 ? &lfx::ring_unit_org_kernel<V, 3, false, true> : chunks == 6 ? &lfx::ring_unit_org_kernel<V, 6, false, true> :
      &lfx::ring_unit_org_kernel<V, lfx::kUnitMaxChunks, false, true>;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * lfx::kUnitWaves), lds_pad, st,
    a.prm, a.cap, a.flags, a.max_rings, a.drop_zero, a.pts, a.scan_begin, a.ring_count, a.tab, a.xform, a.geom);
}

// second: the pass over the rings ring_order_kernel repaired (list = redo list, defer = slow list); else the first pass over
// the scans on the fall-back list, loop = the form that walks a list longer than its grid
void LFX_CAT(launch_unit_v, LFX_VARIANT)(bool second, int chunks, bool loop, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitArgs & a)
{
  constexpr int V = LFX_VARIANT;
  void (*kern)(lfx::Params, uint32_t, uint32_t, uint32_t, const uint32_t *, const float2 *, const float *, const uint32_t *,
    const lfx::UnitTables *, uint32_t *, uint32_t *, const uint32_t *, const uint32_t *, uint32_t) = nullptr;
#define LFX_PICK_UNIT(SECONDV, LOOPV) \
  (chunks == 5 ? &lfx::ring_unit_kernel<V, SECONDV, 5, LOOPV> : chunks == 4 ? &lfx::ring_unit_kernel<V, SECONDV, 4, LOOPV> : \
   chunks == 3 ? &lfx::ring_unit_kernel<V, SECONDV, 3, LOOPV> : chunks == 6 ? &lfx::ring_unit_kernel<V, SECONDV, 6, LOOPV> : \
   &lfx::ring_unit_kernel<V, SECONDV, lfx::kUnitMaxChunks, LOOPV>)
  kern = second ? LFX_PICK_UNIT(true, false) : (loop ? LFX_PICK_UNIT(false, true) : LFX_PICK_UNIT(false, false));
#undef LFX_PICK_UNIT
  hipLaunchKernelGGL(kern, grid, dim3(64 * lfx::kUnitWaves), lds_pad, st,
    a.prm, a.cap, a.flags, a.max_rings, a.ring_count, a.sxy, a.sz, a.sidx, a.tab, a.defer_count, a.defer_list, a.list_count, a.list,
    a.redo_cap);
}

}  // namespace lfx_host

#if defined(LFX_STAMPS) && LFX_VARIANT == 0
// Diagnostic build only: copy the unit kernel's stage stamps (see LFX_STAMP; the default variant's kernels) to the host.
extern "C" int lfx_debug_read_stamps(unsigned long long * out, int n)
{
  const int total = lfx::kStampUnits * lfx::kStampSlots;
  if (!out || n < total) {return -total;}
  if (hipDeviceSynchronize() != hipSuccess) {return LFX_ERR_HIP;}
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lfx::g_unit_stamps), sizeof(unsigned long long) * total) != hipSuccess) {return LFX_ERR_HIP;}
  return total;
}
#endif
