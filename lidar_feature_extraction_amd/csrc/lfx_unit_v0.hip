// lfx_unit_v0.hip -- the unit kernels of parameter variant 0 (lfx_kernels_unit.hpp, UnitVariant)
#define LFX_VARIANT 0
#include "lfx_unit_variant.inl"
