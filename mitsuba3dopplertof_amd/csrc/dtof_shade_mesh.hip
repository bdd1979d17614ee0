// dtof_shade_mesh.hip -- instantiations of k_shade (dtof_shade.h): diffuse scenes with triangle meshes / analytic shapes (MESH = true, SPEC = 0).
#include "dtof_shade.h"

namespace dtof {

void launch_shade_mesh(bool area, bool k4, const ShadeLaunch &L) {
    if (area) { if (k4) launch_shade_variant<true, kMaxOffsets, true, 0>(L); else launch_shade_variant<true, 1, true, 0>(L); }
    else      { if (k4) launch_shade_variant<false, kMaxOffsets, true, 0>(L); else launch_shade_variant<false, 1, true, 0>(L); }
}

}  // namespace dtof
