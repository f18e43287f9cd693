// TSDFfusion.hpp -- drop-in replacement for the reference's include/TSDFfusion.hpp.
//
// The reference class embeds CPython and forwards to the third-party package
// tsdf-fusion-python (ref: src/TSDFfusion.cpp:42-105, src/TSDFfusion.py.in:14-53), which is
// not vendored and is broken as committed.  This class keeps the public
// surface -- TSDFfusion(), ~TSDFfusion(), Integrate(cv::Mat imRGB, cv::Mat imD)
// (ref: include/TSDFfusion.hpp:36-49) -- and backs it with the native HIP library: no
// interpreter, no Python.h.  The volume is the one the Python glue builds
// (ref: src/TSDFfusion.py.in:19-29): bounds [0,10]^3 m in the world frame, 0.02 m voxels
// (500^3), TUM fr3 intrinsics, observation weight 1.
//
// The reference's C++ signature carries no camera pose although its Python side needs one
// (ref: include/TSDFfusion.hpp:49 vs src/TSDFfusion.py.in:31).  Here: SetPose() before the
// 2-argument Integrate (which uses the last pose set, identity at first), or the 3-argument
// overload.  Colour is fused beside the distance as tsdf-fusion-python does it (per channel the
// weighted running mean, rounded and clamped; csrc/tsdf_colour.hip.h) and SaveMesh writes it per
// vertex.  The voxel update is the reference's own GpuIntegrate rule (ref: src/tsdf.cu:15-60); parity
// with tsdf-fusion-python's arithmetic is unpinned because that package is absent.
#ifndef TSDF_HIP_DROPIN_TSDFFUSION_HPP
#define TSDF_HIP_DROPIN_TSDFFUSION_HPP

#include <string>

#include "tsdf_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define TSDFFUSION_HAVE_OPENCV 1
#endif
#endif

class TSDFfusion
{
	public:

		/**	Default constructor (ref: include/TSDFfusion.hpp:36) */
		TSDFfusion();

		/**	Default destructor (ref: include/TSDFfusion.hpp:40) */
		~TSDFfusion();

#ifdef TSDFFUSION_HAVE_OPENCV
		/**	Integrate an RGB-D frame (ref: include/TSDFfusion.hpp:49)
		@param imRGB colour image, CV_8UC3 of the depth image's size, channel 0 into the low byte (may be empty)
		@param imD depth image, CV_32F metres, 480x640 contiguous
		*/
		void Integrate(cv::Mat imRGB, cv::Mat imD);
		void Integrate(cv::Mat imRGB, cv::Mat imD, cv::Mat cam2world);
#endif
		/** OpenCV-free forms: depth is height*width floats in metres, rgb height*width*3 bytes or NULL. */
		void Integrate(const unsigned char *rgb, const float *depth, int height, int width);
		void Integrate(const unsigned char *rgb, const float *depth, int height, int width,
		               const float cam2world[16]);

		/** Pose used by the pose-less Integrate calls (row-major 4x4 camera-to-world). */
		void SetPose(const float cam2world[16]);

		/** Surface points to a .ply (the reference's SaveMesh needs the absent Python package;
		    ref: src/TSDFfusion.py.in:48-53).  Point cloud in the format of ref: src/tsdf.cu:185-212. */
		void SavePointCloud(const std::string &file_name);

		/** Triangle mesh of the fused surface to a binary .ply -- the native stand-in for the Python glue's
		    SaveMesh (ref: src/TSDFfusion.py.in:48-53), by marching tetrahedra on the device. */
		void SaveMesh(const std::string &file_name);

		tsdf_volume *handle() const { return vol_; }

	private:

		TSDFfusion(const TSDFfusion &);
		TSDFfusion &operator=(const TSDFfusion &);
		void initialise();
		tsdf_volume *vol_;
		float pose_[16];
};

#endif // TSDF_HIP_DROPIN_TSDFFUSION_HPP
