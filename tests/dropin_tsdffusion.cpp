// dropin_tsdffusion.cpp -- the reference's second TSDF backend, `class TSDFfusion`
// (ref: include/TSDFfusion.hpp:25-49), used the way Object.cpp would (ref: src/Object.cpp:68,165
// "tsdf = new TSDFfusion(); ... tsdf->Integrate(imRGB, imD);"), through the OpenCV-free overloads of
// include/TSDFfusion.hpp.  Frames come from a file: int32 n, n x { float cam2world[16], float depth[480*640],
// uint8 rgb[480*640*3] }.
// Writes the surface point cloud to the path given as argv[2].
#include <cstdio>
#include <vector>

#include "TSDFfusion.hpp"

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	FILE *fp = std::fopen(argv[1], "rb");
	if (!fp) return 2;
	int n = 0;
	if (std::fread(&n, 4, 1, fp) != 1) return 2;
	TSDFfusion *tsdf = new TSDFfusion();
	std::vector<float> depth(480 * 640);
	std::vector<unsigned char> rgb(480 * 640 * 3, 0);
	for (int k = 0; k < n; ++k) {
		float pose[16];
		if (std::fread(pose, 4, 16, fp) != 16 || std::fread(depth.data(), 4, depth.size(), fp) != depth.size() ||
		    std::fread(rgb.data(), 1, rgb.size(), fp) != rgb.size())
			return 2;
		tsdf->SetPose(pose);                                     // the reference signature has no pose argument
		tsdf->Integrate(rgb.data(), depth.data(), 480, 640);     // == Integrate(cv::Mat imRGB, cv::Mat imD)
	}
	std::fclose(fp);
	tsdf->SavePointCloud(argv[2]);
	if (argc > 3) tsdf->SaveMesh(argv[3]);                       // ref: SaveMesh of src/TSDFfusion.py.in:48-53
	delete tsdf;
	return 0;
}
