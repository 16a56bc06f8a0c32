// lfx_unit_v1.hip -- the unit kernels of parameter variant 1 (lfx_kernels_unit.hpp, UnitVariant)
#define LFX_VARIANT 1
#include "lfx_unit_variant.inl"
