// dropin_multidevice.cpp -- test harness: the reference's call sequence on its TSDF member (ref: src/Object.cpp:67,164,76:
// construct, Integrate per keyframe, delete) through include/tsdf.hpp, once on one device and once with the grid cut
// into three z-slabs by TSDF(cfg, devices).  Reads dims, voxel size, origin and the frames from argv[1]; writes
// tsdf1.{ply,bin} (single) and tsdf2.{ply,bin} (slabs) into the working directory and compares the host mirrors.
#include <cstdio>
#include <cstring>
#include <vector>

#include "tsdf.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *fp = std::fopen(argv[1], "rb");
    if (!fp) return 2;
    int hdr[4];
    float vs, origin[3];
    if (std::fread(hdr, sizeof(int), 4, fp) != 4 || std::fread(&vs, sizeof(float), 1, fp) != 1 ||
        std::fread(origin, sizeof(float), 3, fp) != 3)
        return 2;
    const int n_frames = hdr[3];
    const size_t px = 480 * 640;
    std::vector<std::vector<float> > poses, depths;
    for (int k = 0; k < n_frames; ++k) {
        std::vector<float> p(16), d(px);
        if (std::fread(p.data(), sizeof(float), 16, fp) != 16 || std::fread(d.data(), sizeof(float), px, fp) != px) return 2;
        poses.push_back(p);
        depths.push_back(d);
    }
    std::fclose(fp);

    tsdf_config cfg;
    tsdf_config_default(&cfg, 480, 640);
    cfg.dim_x = hdr[0]; cfg.dim_y = hdr[1]; cfg.dim_z = hdr[2];
    cfg.z_begin = 0; cfg.z_end = hdr[2];
    cfg.voxel_size = vs;
    cfg.trunc_margin = vs * 5;
    std::memcpy(cfg.origin, origin, sizeof origin);

    TSDF::ThrowOnError(false);
    cfg.id = 1;
    TSDF *single = new TSDF(cfg);
    cfg.id = 2;
    std::vector<int> devices(3, 0);          // three slabs; this box has one card
    TSDF *slabs = new TSDF(cfg, devices);
    for (int k = 0; k < n_frames; ++k) {
        single->Integrate(depths[k].data(), poses[k]);
        slabs->Integrate(depths[k].data(), poses[k]);
    }
    single->Download();
    slabs->Download();
    const size_t n = (size_t)hdr[0] * hdr[1] * hdr[2];
    const bool same = std::memcmp(single->voxel_grid_TSDF, slabs->voxel_grid_TSDF, n * sizeof(float)) == 0 &&
                      std::memcmp(single->voxel_grid_weight, slabs->voxel_grid_weight, n * sizeof(float)) == 0;
    std::printf("%s\n", same ? "mirrors identical" : "MIRRORS DIFFER");
    delete single;   // writes tsdf1.ply, tsdf1.bin
    delete slabs;    // writes tsdf2.ply, tsdf2.bin
    return same ? 0 : 1;
}
