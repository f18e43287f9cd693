// tsdf_dropin.cpp -- the C++ host side above the C ABI: class TSDF and class TSDFfusion with the
// reference's public surface (ref: include/tsdf.hpp:22-43, include/TSDFfusion.hpp:25-49),
// forwarding to libtsdf_hip.so.  Built into libtsdf_dropin.so with plain g++.
#include "tsdf.hpp"
#include "TSDFfusion.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <new>
#include <stdexcept>

namespace {
bool g_throw = false;
const long long kEagerMirrorLimit = 1LL << 28;  // host mirrors above 256 Mi voxels are allocated on Download()
}

void TSDF::ThrowOnError(bool on) { g_throw = on; }

// ref: src/tsdf.cu:405-420 -- the reference prints and exits; opt-in exceptions for library use
void TSDF::fail(const char *what, int line) const
{
	std::string msg = std::string(what) + ": " + tsdf_last_error();
	if (g_throw) throw std::runtime_error(msg);
	std::cerr << "TSDF failure at LINE " << line << ": " << msg << std::endl;
	std::cerr << "FatalError. Program Terminated." << std::endl;
	std::exit(EXIT_FAILURE);
}

TSDF::TSDF(int h, int w, int MOid, std::vector<float> base2world_, std::vector<float> origin)
	: voxel_grid_TSDF(NULL), voxel_grid_weight(NULL), vol_(NULL), grp_(NULL), save_on_destroy_(true)
{
	tsdf_config_default(&cfg_, h, w);  // 200^3 @ 4 mm, trunc 20 mm, TUM K (ref: include/tsdf.hpp:63-67,96)
	cfg_.id = MOid;
	for (size_t i = 0; i < 3 && i < origin.size(); ++i) cfg_.origin[i] = origin[i];           // ref: src/tsdf.cu:66-68
	for (size_t i = 0; i < 16 && i < base2world_.size(); ++i) cfg_.base2world[i] = base2world_[i];  // ref: :72
	init();
}

TSDF::TSDF(const tsdf_config &cfg)
	: voxel_grid_TSDF(NULL), voxel_grid_weight(NULL), cfg_(cfg), vol_(NULL), grp_(NULL), save_on_destroy_(true)
{
	init();
}

TSDF::TSDF(const tsdf_config &cfg, const std::vector<int> &devices)
	: voxel_grid_TSDF(NULL), voxel_grid_weight(NULL), cfg_(cfg), vol_(NULL), grp_(NULL), devices_(devices),
	  save_on_destroy_(true)
{
	cfg_.z_begin = 0;
	cfg_.z_end = cfg_.dim_z;
	init();
}

long long TSDF::voxels() const { return grp_ ? tsdf_group_voxels(grp_) : tsdf_slab_voxels(vol_); }

void TSDF::init()
{
	if (!devices_.empty()) {
		std::vector<int32_t> dev(devices_.begin(), devices_.end());
		if (tsdf_group_create(&cfg_, dev.data(), (int32_t)dev.size(), &grp_) != TSDF_OK) fail("tsdf_group_create", __LINE__);
	} else if (tsdf_create(&cfg_, &vol_) != TSDF_OK) {
		fail("tsdf_create", __LINE__);
	}
	const long long n = voxels();
	if (n <= kEagerMirrorLimit) {
		// ref: src/tsdf.cu:77-81 -- host mirrors exist from construction, TSDF = 1, weight = 0
		voxel_grid_TSDF = new float[n > 0 ? n : 1];
		voxel_grid_weight = new float[n > 0 ? n : 1];
		for (long long i = 0; i < n; ++i) voxel_grid_TSDF[i] = 1.0f;
		std::memset(voxel_grid_weight, 0, sizeof(float) * (size_t)n);
	}
}

void TSDF::Integrate(float *depth_im, std::vector<float> cam2world_vec)
{
	float cam2world[16] = {0};
	for (size_t i = 0; i < 16 && i < cam2world_vec.size(); ++i) cam2world[i] = cam2world_vec[i];  // ref: src/tsdf.cu:139
	const int rc = grp_ ? tsdf_group_integrate(grp_, depth_im, cam2world) : tsdf_integrate(vol_, depth_im, cam2world);
	if (rc != TSDF_OK) fail("tsdf_integrate", __LINE__);
}

void TSDF::Sync()
{
	if ((grp_ ? tsdf_group_sync(grp_) : tsdf_sync(vol_)) != TSDF_OK) fail("tsdf_sync", __LINE__);
}

int TSDF::download_mirrors()
{
	const long long n = voxels();
	if (!voxel_grid_TSDF) voxel_grid_TSDF = new float[n > 0 ? n : 1];
	if (!voxel_grid_weight) voxel_grid_weight = new float[n > 0 ? n : 1];
	return grp_ ? tsdf_group_download(grp_, voxel_grid_TSDF, voxel_grid_weight)
	            : tsdf_download(vol_, voxel_grid_TSDF, voxel_grid_weight);
}

void TSDF::Download()
{
	if (download_mirrors() != TSDF_OK) fail("tsdf_download", __LINE__);
}

// A destructor never throws: with ThrowOnError(true) a failure of the final download or of a file is reported on stderr and
// the teardown goes on (an exception out of a destructor is std::terminate); by default it ends the program as every other
// failure does (ref: src/tsdf.cu:405-420, checkCUDA inside the reference's destructor: print, reset, exit).
void TSDF::fail_in_destructor(const char *what, int line) const
{
	if (!g_throw) fail(what, line);
	std::cerr << "TSDF::~TSDF: " << what << " failed at LINE " << line << ": " << tsdf_last_error()
	          << " (tsdf" << cfg_.id << ".ply / .bin may be missing or incomplete)" << std::endl;
}

TSDF::~TSDF()
{
	if (vol_ || grp_) {
		if (save_on_destroy_) {
			// The mirrors are the reference's public members, refreshed here as it does (ref: src/tsdf.cu:101-104); the two files
			// do not depend on them -- the writers stream device memory to the file through pinned staging pieces -- so a grid whose
			// 2 x n host floats cannot be allocated, or whose download fails, still gets the files its destructor exists to write.
			try {
				if (download_mirrors() != TSDF_OK) fail_in_destructor("tsdf_download", __LINE__);
			} catch (const std::bad_alloc &) {   // the lazily allocated mirrors of a large grid
				std::cerr << "TSDF::~TSDF: out of host memory for the mirrors (the files are written all the same)" << std::endl;
			}
			// ref: src/tsdf.cu:109-112 -- surface points, weight threshold 0.9 (tsdf_thresh 1.2 is unused there)
			std::string name = "tsdf" + std::to_string(cfg_.id) + ".ply";
			if ((grp_ ? tsdf_group_save_ply(grp_, name.c_str(), 0.9f) : tsdf_save_ply(vol_, name.c_str(), 0.9f)) != TSDF_OK)
				fail_in_destructor("tsdf_save_ply", __LINE__);
			name = "tsdf" + std::to_string(cfg_.id) + ".bin";  // ref: src/tsdf.cu:116-132
			if ((grp_ ? tsdf_group_save_bin(grp_, name.c_str()) : tsdf_save_bin(vol_, name.c_str())) != TSDF_OK)
				fail_in_destructor("tsdf_save_bin", __LINE__);
		}
		if (grp_) tsdf_group_destroy(grp_);
		if (vol_) tsdf_destroy(vol_);
		vol_ = NULL;
		grp_ = NULL;
	}
	delete[] voxel_grid_TSDF;    // the reference leaks both mirrors and the device buffers
	delete[] voxel_grid_weight;
}

// ------------------------------------------------------------------------------------------------
// TSDFfusion
// ------------------------------------------------------------------------------------------------
TSDFfusion::TSDFfusion() : vol_(NULL)
{
	initialise();
}

void TSDFfusion::initialise()
{
	std::cout << " * Initialising TSDFfusion ... ";  // ref: src/TSDFfusion.cpp:44
	tsdf_config cfg;
	tsdf_config_default(&cfg, 480, 640);
	// ref: src/TSDFfusion.py.in:19-29 -- bounds [0,10]^3, voxel 0.02 -> 500^3; K is the same TUM fr3 matrix
	cfg.dim_x = cfg.dim_y = cfg.dim_z = 500;
	cfg.z_begin = 0;
	cfg.z_end = 500;
	cfg.voxel_size = 0.02f;
	cfg.trunc_margin = cfg.voxel_size * 5;
	cfg.origin[0] = cfg.origin[1] = cfg.origin[2] = 0.0f;
	if (tsdf_create(&cfg, &vol_) != TSDF_OK)
		throw std::runtime_error(std::string("Could not create the TSDF volume: ") + tsdf_last_error());
	// colour beside the distance, as the Python glue's volume keeps it (ref: src/TSDFfusion.py.in:43)
	if (tsdf_colour_enable(vol_) != TSDF_OK)
		throw std::runtime_error(std::string("Could not allocate the colour volume: ") + tsdf_last_error());
	std::memset(pose_, 0, sizeof pose_);
	pose_[0] = pose_[5] = pose_[10] = pose_[15] = 1.0f;
	std::cout << "Done !" << std::endl;
}

TSDFfusion::~TSDFfusion()
{
	if (vol_) tsdf_destroy(vol_);
	std::cout << "TSDFfusion has been deleted." << std::endl;  // ref: src/TSDFfusion.cpp:36
}

void TSDFfusion::SetPose(const float cam2world[16]) { std::memcpy(pose_, cam2world, sizeof pose_); }

void TSDFfusion::Integrate(const unsigned char *rgb, const float *depth, int height, int width, const float cam2world[16])
{
	tsdf_config cfg;
	tsdf_get_config(vol_, &cfg);
	if (height != cfg.im_height || width != cfg.im_width)
		throw std::runtime_error("TSDFfusion::Integrate: depth image must be 480x640");
	// ref: src/TSDFfusion.py.in:43 -- colour and depth fused together; without a colour image only the geometry
	const int rc = rgb ? tsdf_integrate_rgbd(vol_, depth, rgb, cam2world) : tsdf_integrate(vol_, depth, cam2world);
	if (rc != TSDF_OK)
		throw std::runtime_error(std::string("TSDFfusion::Integrate: ") + tsdf_last_error());
}

void TSDFfusion::Integrate(const unsigned char *rgb, const float *depth, int height, int width)
{
	Integrate(rgb, depth, height, width, pose_);
}

#ifdef TSDFFUSION_HAVE_OPENCV
void TSDFfusion::Integrate(cv::Mat imRGB, cv::Mat imD)
{
	if (imD.type() != CV_32F || !imD.isContinuous())
		throw std::runtime_error("TSDFfusion::Integrate: depth must be continuous CV_32F metres");
	if (!imRGB.empty() && (imRGB.type() != CV_8UC3 || !imRGB.isContinuous() || imRGB.rows != imD.rows || imRGB.cols != imD.cols))
		throw std::runtime_error("TSDFfusion::Integrate: colour must be continuous CV_8UC3 of the depth image's size");
	Integrate(imRGB.empty() ? NULL : imRGB.data, (const float *)imD.data, imD.rows, imD.cols, pose_);
}

void TSDFfusion::Integrate(cv::Mat imRGB, cv::Mat imD, cv::Mat cam2world)
{
	cv::Mat p;
	cam2world.convertTo(p, CV_32F);
	p = p.clone();
	SetPose((const float *)p.data);
	Integrate(imRGB, imD);
}
#endif

void TSDFfusion::SavePointCloud(const std::string &file_name)
{
	if (tsdf_save_ply(vol_, file_name.c_str(), 0.9f) != TSDF_OK)
		throw std::runtime_error(std::string("TSDFfusion::SavePointCloud: ") + tsdf_last_error());
}

void TSDFfusion::SaveMesh(const std::string &file_name)
{
	std::cout << "Saving to " << file_name << " ... " << std::endl;  // ref: src/TSDFfusion.py.in:51
	// verts, faces, norms, colors as the Python glue's get_mesh + meshwrite hand them to the file
	if (tsdf_save_mesh_welded_ply(vol_, file_name.c_str(), 0.9f) != TSDF_OK)
		throw std::runtime_error(std::string("TSDFfusion::SaveMesh: ") + tsdf_last_error());
}
