// dropin_callsite.cpp -- replays, against include/tsdf.hpp of THIS repository, exactly the call
// sequence the reference's Object makes on its TSDF member when the commented-out lines are
// restored:
//     Object::Object      tsdf = new TSDF(mnHeight, mnWidth, mnId, base2world, origin);   ref: src/Object.cpp:67
//     Object::Integrate   tsdf->Integrate(depth, cam2world_vec);                          ref: src/Object.cpp:160-164
//     Object::~Object     delete(tsdf);   -> tsdf<id>.ply, tsdf<id>.bin in the CWD         ref: src/Object.cpp:76
// Object.cpp itself cannot be compiled here (needs ORB_SLAM2 + OpenCV), so the frames come from a
// file the test writes:  int32 id, int32 n, float base2world[16], float origin[3],
//                        n x { float cam2world[16], float depth[480*640] }.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "tsdf.hpp"

int main(int argc, char **argv)
{
	if (argc < 2) { std::fprintf(stderr, "usage: %s frames.bin [--throw]\n", argv[0]); return 2; }
	// --throw: the opt-in exception mode; the destructor must report a failed file on stderr and return normally
	// (tests/test_gpu_dropin.py runs this in a directory it cannot write to)
	if (argc > 2 && std::string(argv[2]) == "--throw") TSDF::ThrowOnError(true);
	FILE *fp = std::fopen(argv[1], "rb");
	if (!fp) { std::perror(argv[1]); return 2; }
	int id = 0, n = 0;
	std::vector<float> base2world(16), origin(3);
	if (std::fread(&id, 4, 1, fp) != 1 || std::fread(&n, 4, 1, fp) != 1 ||
	    std::fread(base2world.data(), 4, 16, fp) != 16 || std::fread(origin.data(), 4, 3, fp) != 3) return 2;

	const int mnHeight = 480, mnWidth = 640;  // ref: src/Object.cpp:18
	TSDF *tsdf = new TSDF(mnHeight, mnWidth, id, base2world, origin);

	std::vector<float> depth_store((size_t)mnHeight * mnWidth);
	for (int k = 0; k < n; ++k) {
		float pose[16];
		if (std::fread(pose, 4, 16, fp) != 16 ||
		    std::fread(depth_store.data(), 4, depth_store.size(), fp) != depth_store.size()) return 2;
		float *depth = depth_store.data();                                   // (float*)imD.data
		std::vector<float> cam2world_vec;
		cam2world_vec.insert(cam2world_vec.end(), pose, pose + 16);          // ref: src/Object.cpp:161-162
		tsdf->Integrate(depth, cam2world_vec);
		for (size_t i = 0; i < depth_store.size(); ++i) depth_store[i] = -1.0f;  // the caller's Mat goes away
	}
	std::fclose(fp);

	// public mirrors exist from construction on, as in the reference (ref: include/tsdf.hpp:40-43)
	if (!tsdf->voxel_grid_TSDF || !tsdf->voxel_grid_weight) return 3;
	delete (tsdf);
	std::puts("destructor returned");
	return 0;
}
