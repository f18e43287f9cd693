// dropin_tsdffusion_cv.cpp -- the reference's own call shape on its second backend, `tsdf->Integrate(imRGB, imD)` with
// cv::Mat arguments (ref: include/TSDFfusion.hpp:49, src/Object.cpp:165), through the cv::Mat overloads of
// include/TSDFfusion.hpp.  Built against tests/fake_opencv (a few cv::Mat members: TEST SCAFFOLDING) on boxes without
// OpenCV, together with csrc/tsdf_dropin.cpp so that the overloads exist.  Same input file and outputs as
// dropin_tsdffusion.cpp; the poses go through the 3-argument overload as double-precision 4x4 Mats.
#include <cstdio>
#include <vector>

#include "TSDFfusion.hpp"

#ifndef TSDFFUSION_HAVE_OPENCV
#error "the cv::Mat overloads are not visible: opencv2/core.hpp was not found on the include path"
#endif

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	FILE *fp = std::fopen(argv[1], "rb");
	if (!fp) return 2;
	int n = 0;
	if (std::fread(&n, 4, 1, fp) != 1) return 2;
	TSDFfusion *tsdf = new TSDFfusion();
	for (int k = 0; k < n; ++k) {
		float pose[16];
		cv::Mat imD(480, 640, CV_32F), imRGB(480, 640, CV_8UC3);
		if (std::fread(pose, 4, 16, fp) != 16 || std::fread(imD.data, 4, 480 * 640, fp) != 480 * 640 ||
		    std::fread(imRGB.data, 1, 480 * 640 * 3, fp) != 480 * 640 * 3)
			return 2;
		cv::Mat Twc(4, 4, CV_64F);                      // ORB_SLAM2 poses are CV_32F; a double Mat exercises convertTo
		for (int i = 0; i < 16; ++i) ((double *)Twc.data)[i] = pose[i];
		if (k % 2 == 0) {
			tsdf->Integrate(imRGB, imD, Twc);
		} else {
			tsdf->SetPose(pose);
			tsdf->Integrate(imRGB, imD);                // the reference's 2-argument signature
		}
	}
	std::fclose(fp);
	tsdf->SavePointCloud(argv[2]);
	if (argc > 3) tsdf->SaveMesh(argv[3]);
	delete tsdf;
	return 0;
}
