// BNMTF Gibbs and BNMF VB drivers (C ABI part 2).
#include "model.h"

using namespace bnmtf;

extern "C" {

int bnmtf_set_state(bnmtf_handle, const double*, const double*, const double*, double) { set_error("BNMTF: not built yet"); return BNMTF_EINVAL; }
int bnmtf_get_state(bnmtf_handle, double*, double*, double*, double*) { set_error("BNMTF: not built yet"); return BNMTF_EINVAL; }
int bnmtf_cond_params(bnmtf_handle, int, int, int, double*, double*) { set_error("BNMTF: not built yet"); return BNMTF_EINVAL; }
int bnmtf_gibbs_run(bnmtf_handle, int, int, float*, float*, float*, double*, double*, double*) { set_error("BNMTF: not built yet"); return BNMTF_EINVAL; }
int bnmf_vb_set_state(bnmtf_handle, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double) { set_error("VB: not built yet"); return BNMTF_EINVAL; }
int bnmf_vb_get_state(bnmtf_handle, double*, double*, double*, double*, double*, double*, double*, double*) { set_error("VB: not built yet"); return BNMTF_EINVAL; }
int bnmf_vb_update(bnmtf_handle, int, int, int) { set_error("VB: not built yet"); return BNMTF_EINVAL; }
int bnmf_vb_exp_square_diff(bnmtf_handle, double*) { set_error("VB: not built yet"); return BNMTF_EINVAL; }
int bnmf_vb_run(bnmtf_handle, int, double*, double*, double*, double*) { set_error("VB: not built yet"); return BNMTF_EINVAL; }

}
