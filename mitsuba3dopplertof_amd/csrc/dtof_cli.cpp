// dtof-render -- native command line front end over the C ABI (include/dtof.h), the counterpart of
// `mitsuba scene.xml -m <variant> -D key=value -o out` (src/mitsuba/mitsuba.cpp:150-423) for the plugins libdtof implements.
//
//   dtof-render scene.xml [-D key=value ...] [-o out.npy|out.pfm] [--spp N] [--seed S] [-m hip_rgb]
//
// Exit code -1 and "Error: ..." on stderr when loading or rendering fails (mitsuba.cpp:366-397,423).
#include "../../include/dtof.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool write_npy(const char *path, const float *img, int h, int w) {
    FILE *f = fopen(path, "wb"); if (!f) return false;
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(h) + ", " + std::to_string(w) + ", 3), }";
    size_t total = 10 + dict.size() + 1, pad = (64 - total % 64) % 64;
    dict += std::string(pad, ' ') + "\n";
    unsigned short hl = (unsigned short) dict.size();
    fwrite("\x93NUMPY\x01\x00", 1, 8, f); fwrite(&hl, 2, 1, f); fwrite(dict.data(), 1, dict.size(), f);
    fwrite(img, 4, (size_t) h * w * 3, f); fclose(f); return true;
}
static bool write_pfm(const char *path, const float *img, int h, int w) {
    FILE *f = fopen(path, "wb"); if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    for (int y = h - 1; y >= 0; --y) fwrite(img + (size_t) y * w * 3, 4, (size_t) w * 3, f);
    fclose(f); return true;
}

int main(int argc, char **argv) {
    std::string scene, out; std::vector<std::string> names, values; unsigned spp = 0, seed = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char *what) -> std::string { if (i + 1 >= argc) { fprintf(stderr, "Error: %s expects a value\n", what); exit(-1); } return argv[++i]; };
        if (a == "-D") { std::string kv = next("-D"); size_t e = kv.find('='); if (e == std::string::npos) { fprintf(stderr, "Error: -D expects key=value\n"); return -1; }
                         names.push_back(kv.substr(0, e)); values.push_back(kv.substr(e + 1)); }
        else if (a.rfind("-D", 0) == 0 && a.size() > 2) { std::string kv = a.substr(2); size_t e = kv.find('='); if (e == std::string::npos) { fprintf(stderr, "Error: -D expects key=value\n"); return -1; }
                         names.push_back(kv.substr(0, e)); values.push_back(kv.substr(e + 1)); }
        else if (a == "-o") out = next("-o");
        else if (a == "--spp") spp = (unsigned) atoi(next("--spp").c_str());
        else if (a == "--seed") seed = (unsigned) atoi(next("--seed").c_str());
        else if (a == "-m") (void) next("-m");   // variant: only hip_rgb exists
        else if (a == "-h" || a == "--help") { printf("usage: dtof-render scene.xml [-D key=value ...] [-o out.npy|out.pfm] [--spp N] [--seed S]\n%s\n", dtof_version()); return 0; }
        else if (scene.empty()) scene = a;
        else { fprintf(stderr, "Error: unexpected argument \"%s\"\n", a.c_str()); return -1; }
    }
    if (scene.empty()) { fprintf(stderr, "Error: no scene file given\n"); return -1; }
    if (out.empty()) out = scene.substr(0, scene.rfind('.')) + ".npy";
    std::vector<const char *> n, v; for (auto &x : names) n.push_back(x.c_str()); for (auto &x : values) v.push_back(x.c_str());
    dtof_scene *sc = nullptr;
    if (dtof_scene_load_file(scene.c_str(), n.data(), v.data(), (int) n.size(), &sc)) { fprintf(stderr, "Error: %s\n", dtof_last_error()); return -1; }
    dtof_scene_info info; dtof_scene_get_info(sc, &info);
    std::vector<float> img((size_t) info.crop_width * info.crop_height * 3);
    dtof_render_stats st;
    if (dtof_render(sc, 0, seed, spp, img.data(), &st)) { fprintf(stderr, "Error: %s\n", dtof_last_error()); dtof_scene_destroy(sc); return -1; }
    bool pfm = out.size() > 4 && out.substr(out.size() - 4) == ".pfm";
    bool ok = pfm ? write_pfm(out.c_str(), img.data(), info.crop_height, info.crop_width) : write_npy(out.c_str(), img.data(), info.crop_height, info.crop_width);
    if (!ok) { fprintf(stderr, "Error: could not write \"%s\"\n", out.c_str()); dtof_scene_destroy(sc); return -1; }
    fprintf(stderr, "Rendering finished. (%dx%d, %llu paths, %.2f ms on the GPU, %.0f Mpaths/s) -> %s\n", info.crop_width, info.crop_height,
            (unsigned long long) st.n_paths, st.ms_total, st.n_paths / (st.ms_total * 1e3), out.c_str());
    dtof_scene_destroy(sc);
    return 0;
}
