// dtof-render -- native command line front end over the C ABI (include/dtof.h), the counterpart of
// `mitsuba scene.xml -m <variant> -D key=value -o out` (src/mitsuba/mitsuba.cpp:150-423) for the plugins libdtof implements.
//
//   dtof-render scene.xml [-D key=value ...] [-o out.npy|out.pfm] [--spp N] [--seed S] [-m hip_rgb] [--gpus G [--stripes ROWS]]
//
// --gpus G: one host thread per GPU of this node, each with its own scene handle and its own RCCL communicator (ncclCommInitAll); thread g
// renders the interleaved stripes of pixel rows g owns (dtof_render_stripes) into a film on its device, ONE ncclReduce(sum) over xGMI brings
// the films to GPU 0, which develops (RGB / W) and downloads the image -- the film never touches host memory.  The torch.distributed
// launcher (python -m mitsuba3dopplertof_amd under torch.distributed.run) is the same exchange with one process per GPU.
// DTOF_CLI_SHARE_GPU=1 (development on one-GPU boxes): all shards run on GPU 0 and are summed on the host, RCCL cannot place two ranks on one device.
//
// Exit code -1 and "Error: ..." on stderr when loading or rendering fails (mitsuba.cpp:366-397,423).
#include "../../include/dtof.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <atomic>

static bool write_npy(const char *path, const float *img, int h, int w, int ch) {
    FILE *f = fopen(path, "wb"); if (!f) return false;
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(h) + ", " + std::to_string(w) + ", " + std::to_string(ch) + "), }";
    size_t total = 10 + dict.size() + 1, pad = (64 - total % 64) % 64;
    dict += std::string(pad, ' ') + "\n";
    unsigned short hl = (unsigned short) dict.size();
    fwrite("\x93NUMPY\x01\x00", 1, 8, f); fwrite(&hl, 2, 1, f); fwrite(dict.data(), 1, dict.size(), f);
    fwrite(img, 4, (size_t) h * w * ch, f); fclose(f); return true;
}
static bool write_pfm(const char *path, const float *img, int h, int w, int ch) {   // PFM has three channels: the alpha of an rgba film is dropped
    FILE *f = fopen(path, "wb"); if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    for (int y = h - 1; y >= 0; --y)
        for (int x = 0; x < w; ++x) fwrite(img + ((size_t) y * w + x) * ch, 4, 3, f);
    fclose(f); return true;
}

int main(int argc, char **argv) {
    std::string scene, out; std::vector<std::string> names, values; unsigned spp = 0, seed = 0; int gpus = 1, stripes = 4;
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
        else if (a == "--gpus") gpus = atoi(next("--gpus").c_str());
        else if (a == "--stripes") stripes = atoi(next("--stripes").c_str());
        else if (a == "-m") (void) next("-m");   // variant: only hip_rgb exists
        else if (a == "-h" || a == "--help") { printf("usage: dtof-render scene.xml [-D key=value ...] [-o out.npy|out.pfm] [--spp N] [--seed S] [--gpus G [--stripes ROWS]]\n%s\n", dtof_version()); return 0; }
        else if (scene.empty()) scene = a;
        else { fprintf(stderr, "Error: unexpected argument \"%s\"\n", a.c_str()); return -1; }
    }
    if (scene.empty()) { fprintf(stderr, "Error: no scene file given\n"); return -1; }
    if (out.empty()) out = scene.substr(0, scene.rfind('.')) + ".npy";
    std::vector<const char *> n, v; for (auto &x : names) n.push_back(x.c_str()); for (auto &x : values) v.push_back(x.c_str());
    dtof_scene *sc = nullptr;
    if (dtof_scene_load_file(scene.c_str(), n.data(), v.data(), (int) n.size(), &sc)) { fprintf(stderr, "Error: %s\n", dtof_last_error()); return -1; }
    dtof_scene_info info; dtof_scene_get_info(sc, &info);
    // hdrfilm pixel_format = rgba: four channels out, and the device films carry the alpha film as a second RGBW plane (include/dtof.h, dtof_scene_set_film_layout)
    const int ch = info.has_alpha ? 4 : 3, planes = info.has_alpha ? 2 : 1;
    std::vector<float> img((size_t) info.crop_width * info.crop_height * ch);
    dtof_render_stats st;
    const bool force_collective = getenv("DTOF_CLI_FORCE_RCCL") != nullptr;   // take the multi-GPU path with --gpus 1 too (a one-rank communicator)
    if (gpus > 1 || force_collective) {
        int visible = 0; (void) hipGetDeviceCount(&visible);
        const bool share = getenv("DTOF_CLI_SHARE_GPU") != nullptr;   // development: all shards on GPU 0 (one-GPU boxes), host sum
        if (gpus > visible && !share) { fprintf(stderr, "Error: --gpus %d but only %d GPU(s) are visible\n", gpus, visible); dtof_scene_destroy(sc); return -1; }
        if (stripes <= 0) { fprintf(stderr, "Error: --stripes expects a positive number of rows\n"); dtof_scene_destroy(sc); return -1; }
        const size_t n_pixels = (size_t) info.crop_width * info.crop_height, film_floats = n_pixels * 4 * planes;
        std::vector<ncclComm_t> comms(gpus, nullptr);
        if (!share) {
            std::vector<int> devs(gpus); for (int g = 0; g < gpus; ++g) devs[g] = g;
            const ncclResult_t rc = ncclCommInitAll(comms.data(), gpus, devs.data());
            if (rc != ncclSuccess) { fprintf(stderr, "Error: ncclCommInitAll: %s\n", ncclGetErrorString(rc)); dtof_scene_destroy(sc); return -1; }
        }
        std::vector<std::vector<float>> films(share ? gpus : 0, std::vector<float>(film_floats));
        std::vector<std::string> errors(gpus); std::vector<dtof_render_stats> stats(gpus);
        std::vector<std::thread> workers;
        // The collective is entered by every rank or by none: each worker finishes what can fail before it (device, scene, film, stream, render), all
        // meet at a host barrier, and the reduce runs only if nobody reported an error -- a rank that stayed away would leave the others waiting in
        // ncclReduce forever, and the image of a failed render is discarded anyway.
        std::mutex gate_mutex; std::condition_variable gate_cv; int arrived = 0; std::atomic<int> failed { 0 };
        auto meet = [&] { std::unique_lock<std::mutex> lock(gate_mutex); if (++arrived == gpus) gate_cv.notify_all(); else gate_cv.wait(lock, [&] { return arrived == gpus; }); };
        for (int g = 0; g < gpus; ++g) workers.emplace_back([&, g] {
            dtof_scene *mine = nullptr; float *d_film = nullptr, *d_rgb = nullptr; hipStream_t stream = nullptr;
            if (hipSetDevice(share ? 0 : g) != hipSuccess) errors[g] = "hipSetDevice failed";
            else if (dtof_scene_load_file(scene.c_str(), n.data(), v.data(), (int) n.size(), &mine)) errors[g] = dtof_last_error();
            else if (hipMalloc((void **) &d_film, film_floats * 4) != hipSuccess || hipMemset(d_film, 0, film_floats * 4) != hipSuccess) errors[g] = "device film allocation failed";
            else if (!share && hipStreamCreate(&stream) != hipSuccess) errors[g] = "stream creation failed";
            else if (dtof_scene_set_film_layout(mine, planes, 0)) errors[g] = dtof_last_error();
            else if (dtof_render_stripes(mine, seed, spp, g * stripes, stripes, gpus * stripes, nullptr, 0, d_film, &stats[g])) errors[g] = dtof_last_error();
            else if (hipDeviceSynchronize() != hipSuccess) errors[g] = "device synchronisation failed";   // the library renders on its own stream
            if (!errors[g].empty()) failed.fetch_add(1);
            meet();
            if (failed.load() == 0) {
                if (share) {
                    if (hipMemcpy(films[g].data(), d_film, film_floats * 4, hipMemcpyDeviceToHost) != hipSuccess) errors[g] = "film download failed";
                } else {
                    const ncclResult_t rc = ncclReduce(d_film, d_film, film_floats, ncclFloat, ncclSum, 0, comms[g], stream);
                    if (rc != ncclSuccess) errors[g] = std::string("ncclReduce: ") + ncclGetErrorString(rc);
                    else if (hipStreamSynchronize(stream) != hipSuccess) errors[g] = "film reduce failed";
                    if (g == 0 && errors[g].empty()) {   // HDRFilm::develop (hdrfilm.cpp:305-406) of the summed film, on the device
                        if (hipMalloc((void **) &d_rgb, n_pixels * 4 * ch) != hipSuccess) errors[g] = "image allocation failed";
                        else if (info.has_alpha ? dtof_develop_rgba(d_film, d_film + n_pixels * 4, d_rgb, (int64_t) n_pixels) : dtof_develop(d_film, d_rgb, (int64_t) n_pixels)) errors[g] = dtof_last_error();
                        else if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(img.data(), d_rgb, n_pixels * 4 * ch, hipMemcpyDeviceToHost) != hipSuccess) errors[g] = "image download failed";
                    }
                }
            }
            if (stream) (void) hipStreamDestroy(stream);
            if (d_rgb) (void) hipFree(d_rgb);
            if (d_film) (void) hipFree(d_film);
            if (mine) dtof_scene_destroy(mine);
        });
        for (auto &w : workers) w.join();
        for (auto &c : comms) if (c) (void) ncclCommDestroy(c);
        for (int g = 0; g < gpus; ++g) if (!errors[g].empty()) { fprintf(stderr, "Error: GPU %d: %s\n", g, errors[g].c_str()); dtof_scene_destroy(sc); return -1; }
        st = stats[0];
        for (int g = 1; g < gpus; ++g) { st.n_paths += stats[g].n_paths; if (stats[g].ms_total > st.ms_total) st.ms_total = stats[g].ms_total; }
        if (share) for (size_t p = 0; p < n_pixels; ++p) {   // development mode: develop the host sum
            float r = 0.f, gch = 0.f, b = 0.f, wgt = 0.f, a = 0.f, awgt = 0.f;
            for (int g = 0; g < gpus; ++g) {
                const float *f = films[g].data() + 4 * p; r += f[0]; gch += f[1]; b += f[2]; wgt += f[3];
                if (info.has_alpha) { const float *fa = films[g].data() + 4 * (n_pixels + p); a += fa[0]; awgt += fa[3]; }
            }
            if (wgt == 0.f) wgt = 1.f;
            img[ch * p] = r / wgt; img[ch * p + 1] = gch / wgt; img[ch * p + 2] = b / wgt;
            if (info.has_alpha) img[ch * p + 3] = a / (awgt == 0.f ? 1.f : awgt);
        }
    } else
    if (dtof_render(sc, 0, seed, spp, img.data(), &st)) { fprintf(stderr, "Error: %s\n", dtof_last_error()); dtof_scene_destroy(sc); return -1; }
    bool pfm = out.size() > 4 && out.substr(out.size() - 4) == ".pfm";
    bool ok = pfm ? write_pfm(out.c_str(), img.data(), info.crop_height, info.crop_width, ch) : write_npy(out.c_str(), img.data(), info.crop_height, info.crop_width, ch);
    if (!ok) { fprintf(stderr, "Error: could not write \"%s\"\n", out.c_str()); dtof_scene_destroy(sc); return -1; }
    fprintf(stderr, "Rendering finished. (%dx%d, %llu paths, %.2f ms on the GPU, %.0f Mpaths/s) -> %s\n", info.crop_width, info.crop_height,
            (unsigned long long) st.n_paths, st.ms_total, st.n_paths / (st.ms_total * 1e3), out.c_str());
    dtof_scene_destroy(sc);
    return 0;
}
