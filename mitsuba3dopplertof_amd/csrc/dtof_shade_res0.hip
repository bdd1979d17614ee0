// dtof_shade_res0.hip -- instantiations of k_shade (dtof_shade.h): the resident first-bounce kernel (Domino: TLAS and small records in LDS, one persistent block per CU), diffuse scenes.
#include "dtof_shade.h"

namespace dtof {

void launch_shade_resident0(bool area, bool k4, const ShadeLaunch &L) {
    if (area) { if (k4) launch_resident_variant<true, kMaxOffsets, 0>(L); else launch_resident_variant<true, 1, 0>(L); }
    else      { if (k4) launch_resident_variant<false, kMaxOffsets, 0>(L); else launch_resident_variant<false, 1, 0>(L); }
}

}  // namespace dtof
