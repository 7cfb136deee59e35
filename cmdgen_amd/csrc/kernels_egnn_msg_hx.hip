// kernels_egnn_msg_hx.hip - the widths other than 256 of kernels_egnn_msg.hip as a translation unit of their own (build time: __graft_entry__.py)
#define CMDGEN_H_PART 1
#include "kernels_egnn_msg.hip"
