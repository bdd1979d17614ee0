// dtof_shade_plain.hip -- instantiations of k_shade (dtof_shade.h): rectangle-only diffuse scenes (MESH = false, SPEC = 0): the headline kernels of the Cornell-wall benchmark.
#include "dtof_shade.h"

namespace dtof {

void launch_shade_plain(bool area, bool k4, const ShadeLaunch &L) {
    if (area) { if (k4) launch_shade_variant<true, kMaxOffsets, false, 0>(L); else launch_shade_variant<true, 1, false, 0>(L); }
    else      { if (k4) launch_shade_variant<false, kMaxOffsets, false, 0>(L); else launch_shade_variant<false, 1, false, 0>(L); }
}

}  // namespace dtof
