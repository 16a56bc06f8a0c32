// lfx_unit_v3.hip -- the unit kernels of parameter variant 3 (lfx_kernels_unit.hpp, UnitVariant)
#define LFX_VARIANT 3
#include "lfx_unit_variant.inl"
