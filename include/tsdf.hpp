// tsdf.hpp -- drop-in replacement for the reference's include/tsdf.hpp (class TSDF).
//
// Same class name, same three public methods, same two public data members as
// ref: include/tsdf.hpp:22-43, so the call sites in ref: src/Object.cpp:67-68,76,164 compile and
// behave unchanged:
//     tsdf = new TSDF(mnHeight, mnWidth, mnId, base2world, origin);
//     tsdf->Integrate(depth, cam2world_vec);
//     delete(tsdf);          // writes tsdf<id>.ply and tsdf<id>.bin in the working directory
// Unlike the reference header this one needs neither cuda_runtime.h nor OpenCV
// (ref: include/tsdf.hpp:8,15-16): it is plain C++11 over the C ABI in tsdf_hip.h, and the
// arithmetic runs in hand-written HIP kernels on an MI355X (libtsdf_hip.so).
//
// Additions the reference lacks (all optional; defaults reproduce the reference):
//     TSDF(const tsdf_config&)   run-time grid size / voxel size / intrinsics / z-slab / device
//     TSDF(const tsdf_config&, devices)   the same grid cut into z-slabs over several GPUs of the node, in this
//                                process (tsdf_group_*): Integrate fans the frame out, the destructor gathers
//     Download(), Sync()         read results back without destroying the object
//     SetSaveOnDestroy(false)    skip the two files the destructor writes
//     TSDF::ThrowOnError(true)   throw std::runtime_error instead of print + exit(1)
#ifndef TSDF_HIP_DROPIN_TSDF_HPP
#define TSDF_HIP_DROPIN_TSDF_HPP

#include <string>
#include <vector>

#include "tsdf_hip.h"

class TSDF
{
	public:

	/** Object TSDF constructor (ref: include/tsdf.hpp:30, src/tsdf.cu:62-96)
	@param h depth image height
	@param w depth image width
	@param id object id, names the files written at destruction
	@param base2world_vec 16 floats, row-major pose of the base camera (first keyframe's Twc)
	@param origin 3 floats, grid origin in the base camera frame
	Grid: the reference's compile-time 200^3 voxels of 4 mm, truncation 20 mm, TUM fr3 intrinsics.
	*/
	TSDF(int h, int w, int id, std::vector<float> base2world_vec, std::vector<float> origin);

	/** Run-time configured volume (not in the reference). */
	explicit TSDF(const tsdf_config &cfg);

	/** The grid of cfg cut into devices.size() contiguous z-slabs, slab i on HIP device devices[i] (not in the
	reference, which is single-GPU).  Same Integrate / Download / destructor behaviour; results and files are
	bit-identical to the single-device object. */
	TSDF(const tsdf_config &cfg, const std::vector<int> &devices);

	/** Downloads the grid, writes tsdf<id>.ply and tsdf<id>.bin (ref: src/tsdf.cu:98-133). */
	~TSDF();

	/** Integrate one depth image (ref: include/tsdf.hpp:37, src/tsdf.cu:135-168)
	@param depth_im pointer to h*w floats, metres, row-major; borrowed for the call only
	@param cam2world_vec 16 floats, row-major camera pose
	*/
	void Integrate(float *depth_im, std::vector<float> cam2world_vec);

	/// pointer to a TSDF voxel grid (host mirror; refreshed by Download() and by the destructor)
	float * voxel_grid_TSDF;

	/// pointer to a TSDF voxel grid weights (host mirror)
	float * voxel_grid_weight;

	// ---- additions -------------------------------------------------------------------------
	void Download();                        ///< refresh the two host mirrors now
	void Sync();                            ///< wait for queued integrations
	void SetSaveOnDestroy(bool on) { save_on_destroy_ = on; }
	tsdf_volume *handle() const { return vol_; }   ///< NULL for a multi-device object
	tsdf_group *group() const { return grp_; }     ///< NULL for a single-device object
	const tsdf_config &config() const { return cfg_; }
	static void ThrowOnError(bool on);      ///< default false: print to stderr and exit(EXIT_FAILURE)

	private:

	TSDF(const TSDF &);             // one object owns one device volume (the reference never copies it)
	TSDF &operator=(const TSDF &);
	void init();
	int download_mirrors();
	void fail(const char *what, int line) const;
	void fail_in_destructor(const char *what, int line) const;   // never throws (ThrowOnError: prints and goes on)
	long long voxels() const;
	tsdf_config cfg_;
	tsdf_volume *vol_;
	tsdf_group *grp_;
	std::vector<int> devices_;
	bool save_on_destroy_;
};
#endif // TSDF_HIP_DROPIN_TSDF_HPP
