/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" door onto the REAL reference dynamics.  The reference's
 * bbdynamics.cpp declares its functions `static` and is meant to be textually
 * included into one translation unit (bbdynamicswrap.pyx:21 does exactly that),
 * so this file includes it FROM WHERE IT LIES (-I/root/reference/dpilqr, see
 * the Makefile) and exposes the same three operations the Cython wrapper does
 * (bbdynamicswrap.pyx:61-164).  No reference source is copied into the repo;
 * the build output goes to oracle/_ref/ (git-ignored).
 */
#include <cstddef>
#include <cstdint>

#include "bbdynamics.cpp" /* resolved through -I/root/reference/dpilqr */

namespace {
/* Model enum order of bbdynamicswrap.pyx:8-16 */
f_ptr pick_f(int model)
{
    switch (model) {
    case 0: return f_double_int_4d;
    case 1: return f_double_int_6d;
    case 2: return f_car_3d;
    case 3: return f_unicycle_4d;
    case 4: return f_quad_6d;
    case 5: return f_human_6d;
    case 6: return f_human_lin_6d;
    case 7: return f_quad_12d;
    default: return nullptr;
    }
}
const int kNs[8] = {4, 6, 3, 4, 6, 6, 6, 12};
}  // namespace

extern "C" int ref_model_f(int model, double *x, double *u, double *x_dot)
{
    f_ptr fn = pick_f(model);
    if (!fn) return -1;
    fn(x, u, x_dot);
    return 0;
}

extern "C" int ref_model_integrate(int model, double *x, double *u, double dt, double *x_new)
{
    f_ptr fn = pick_f(model);
    if (!fn) return -1;
    rk4(fn, dt, x, u, (size_t)kNs[model], x_new);
    return 0;
}

extern "C" int ref_model_linearize(int model, double *x, double *u, double dt, double *A, double *B)
{
    switch (model) {
    case 0: linearize_double_int_4d(dt, A, B); return 0;
    case 1: linearize_double_int_6d(dt, A, B); return 0;
    case 2: linearize_car_3d(x, u, dt, A, B); return 0;
    case 3: linearize_unicycle_4d(x, u, dt, A, B); return 0;
    case 4: linearize_quad_6d(x, u, dt, A, B); return 0;
    case 5: linearize_human_6d(x, u, dt, A, B); return 0;
    case 6: linearize_human_lin_6d(dt, A, B); return 0;
    case 7: linearize_quad_12d(x, u, dt, A, B); return 0;
    default: return -1;
    }
}
