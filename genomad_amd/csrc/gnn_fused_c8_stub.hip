// What the default build links instead of gnn_fused_c8.hip.  GNN_PREC_F16C8 - one f16 MFMA pass + MX-fp8 (e4m3)
// correction MFMAs, 2.0 pass equivalents - is EXPERIMENTAL: it leaves the 1e-4 tolerance on a few of 10^6 windows
// (profiles/history/r02c6_tails.txt) and cannot ship.  `GNN_EXPERIMENTAL=1 genomad_amd/csrc/build.sh` links the real kernel; this
// file keeps the enum value and the entry points answering with an error instead of a missing symbol.
#include "gnn_common.h"

namespace gnn {

int pack_fused_c8_weights(gnn_ctx*, const gnn_weights*) { return GNN_OK; }       // nothing to pack

int launch_front_c8(gnn_ctx*, const uint8_t*, int64_t) {
    set_error("GNN_PREC_F16C8 is experimental and not part of this build (rebuild with GNN_EXPERIMENTAL=1 genomad_amd/csrc/build.sh)");
    return GNN_ERR_STATE;
}

}  // namespace gnn

extern "C" int gnn_has_experimental(void) { return 0; }
