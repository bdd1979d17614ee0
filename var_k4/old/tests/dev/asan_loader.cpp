// Host-side loader under AddressSanitizer + UBSan (tools/sanitize_loader.sh): XML -> HostScene -> scene blob for every file given;
// a file may be rejected (exception) but must not trip a sanitizer.
#include "dtof_scene.h"
#include <cstdio>
#include <stdexcept>
int main(int argc, char **argv) {
    int ok = 0, err = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            std::string p = argv[i], dir = p.substr(0, p.find_last_of('/'));
            dtof::HostScene sc = dtof::load_scene_xml(dtof::read_file(p), {{"resx", "16"}, {"resy", "16"}}, dir);
            auto blob = dtof::build_scene_blob(sc);
            ok += blob.size() > 0;
        } catch (const std::exception &e) { ++err; }
    }
    printf("ok %d err %d\n", ok, err);
    return 0;
}
