// lfx_unit_v2.hip -- the unit kernels of parameter variant 2 (lfx_kernels_unit.hpp, UnitVariant)
#define LFX_VARIANT 2
#include "lfx_unit_variant.inl"
