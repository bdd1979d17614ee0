// dtof_shade_res2.hip -- instantiations of k_shade (dtof_shade.h): the resident first-bounce kernel, every BSDF + blendbsdf.
#include "dtof_shade.h"

namespace dtof {

void launch_shade_resident2(bool k4, const ShadeLaunch &L) {
    if (k4) launch_resident_variant<true, kMaxOffsets, 2>(L); else launch_resident_variant<true, 1, 2>(L);
}

}  // namespace dtof
