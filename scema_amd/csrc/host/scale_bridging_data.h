// scale_bridging_data.h -- FE <-> MD wire record (reference headers/scale_bridging_data.h:12-24):
// 3 x int32 + padding + 6 f64 strain + 6 f64 stress = 112 bytes, sent as raw bytes (MPI_QP).
#pragma once
#include <vector>

#include "../../../include/scema_stmd.h"

namespace scema {
using QP = scema_qp;
static_assert(sizeof(QP) == 112, "QP must match the reference's 112-byte MPI_QP record");
struct ScaleBridgingData {
  std::vector<QP> update_list;
};
}  // namespace scema
