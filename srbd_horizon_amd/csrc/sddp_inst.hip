// sddp_inst.hip -- one model build of the library: compiled once per entry of srbd_horizon_amd/_lib.py INSTANCES with
//   -DSDDP_INST_MODEL=<device model type>  -DSDDP_INST_FN=<name of the accessor>  -DSDDP_INST_NAME="<model name>"
// (in parallel: the solve kernels of one model build are 10-30 s of device code generation each).  The accessor returns the
// build's table of launchers (sddp_handle.hpp ModelOps); sddp_api.hip picks a table by (model_id, barrier, second_order).
#include "sddp_launch.hpp"

#if !defined(SDDP_INST_MODEL) || !defined(SDDP_INST_FN) || !defined(SDDP_INST_NAME)
#error "compile with -DSDDP_INST_MODEL=... -DSDDP_INST_FN=... -DSDDP_INST_NAME=..."
#endif

namespace sddp {
const ModelOps* SDDP_INST_FN() {
    static const ModelOps ops = make_ops<SDDP_INST_MODEL>(SDDP_INST_NAME);
    return &ops;
}
}  // namespace sddp
