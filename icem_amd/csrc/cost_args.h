// cost_args.h -- icem_cost_spec + icem_cost_terms (include/icem_hip.h) in kernel-argument form.
#pragma once
#include "../../include/icem_hip.h"

namespace icem {

template <typename T>
struct CostArgs {
    T ctrl_w, lin_w, flip_pen, flip_th;
    int lin_idx, flip_idx;
    // icem_cost_terms (include/icem_hip.h); ext = any of them on
    T diff_w, health_pen, health_lo, health_hi, box_lo, box_hi;
    int ext, diff_idx, health_idx, health_closed, box_from, n_terms;
    struct Term {
        T w, th, gate_th;
        int kind, a, b, len, gate_idx;
    } terms[ICEM_MAX_COST_TERMS];
};

}  // namespace icem
