// kernels.hpp -- all hand-written HIP kernels (gfx950 / CDNA4) of the multifrontal LU backend.
// The roofline that bounds each kernel and its algorithmic bytes / flops are stated in DESIGN.md.
#pragma once
#include "kernels_assembly.hpp"
#include "kernels_common.hpp"
#include "kernels_factor.hpp"
#include "kernels_extend_add.hpp"
#include "kernels_factor_front.hpp"
#include "kernels_factor_binv.hpp"
#include "kernels_solve.hpp"
#include "kernels_solve_fused.hpp"
#include "kernels_factor_chain.hpp"
#include "kernels_solve_tree.hpp"
#include "kernels_solve_leaf.hpp"
#include "kernels_vector.hpp"
