// g++ -O2 -o host_copy host_copy.cpp && ./host_copy : one host thread copying 1.2 MB frames into a ring, memcpy against csrc/host_copy.h
#include "../../semantic_slam_amd/csrc/host_copy.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(){ size_t n=1228800; std::vector<std::vector<float>> src(64, std::vector<float>(n/4, 1.5f)); void* dst; posix_memalign(&dst,4096,n*4);
 for(int mode=0;mode<2;++mode){ auto t0=std::chrono::steady_clock::now(); for(int k=0;k<2000;++k){ char* d=(char*)dst+(k%4)*n; if(mode) tsdf_host::copy_to_pinned(d,src[k%64].data(),n); else memcpy(d,src[k%64].data(),n);} auto t1=std::chrono::steady_clock::now(); double us=std::chrono::duration<double,std::micro>(t1-t0).count()/2000; printf("%s: %.1f us per 1.2 MB frame (%.1f GB/s)\n", mode?"streaming":"memcpy", us, n/us/1e3);} }
