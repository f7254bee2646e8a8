// placeholder, replaced by the fused MFMA front end
#include "gnn_common.h"
namespace gnn {
int pack_fused_weights(gnn_ctx*, const gnn_weights*, const std::vector<float>*) { return GNN_OK; }
int launch_front_fused(gnn_ctx*, const uint8_t*, int64_t, int) {
    set_error("fused front end not built");
    return GNN_ERR_STATE;
}
}  // namespace gnn
