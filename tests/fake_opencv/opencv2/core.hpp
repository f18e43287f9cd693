// TEST SCAFFOLDING, not OpenCV: the handful of cv::Mat members that include/TSDFfusion.hpp's cv::Mat overloads touch,
// so that those overloads are compiled and exercised on a box without OpenCV (tests/test_gpu_dropin.py).  The real
// header takes its place wherever OpenCV is installed (TSDFfusion.hpp picks it up through __has_include).
#pragma once
#include <cstddef>
#include <cstring>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)

namespace cv {

class Mat
{
	public:
		Mat() : rows(0), cols(0), data(NULL), type_(0) {}
		Mat(int r, int c, int type) : rows(r), cols(c), type_(type) { store_.resize((size_t)r * c * elem()); data = store_.data(); }
		Mat(int r, int c, int type, void *ext) : rows(r), cols(c), data((unsigned char *)ext), type_(type) {}
		Mat(const Mat &o) : rows(o.rows), cols(o.cols), type_(o.type_), store_(o.store_) { data = store_.empty() ? o.data : store_.data(); }
		Mat &operator=(const Mat &o)
		{
			rows = o.rows; cols = o.cols; type_ = o.type_; store_ = o.store_;
			data = store_.empty() ? o.data : store_.data();
			return *this;
		}
		int type() const { return type_; }
		bool empty() const { return data == NULL || rows * cols == 0; }
		bool isContinuous() const { return true; }
		Mat clone() const
		{
			Mat m(rows, cols, type_);
			std::memcpy(m.data, data, (size_t)rows * cols * elem());
			return m;
		}
		void convertTo(Mat &dst, int rtype) const
		{
			Mat m(rows, cols, rtype);
			const size_t n = (size_t)rows * cols;
			for (size_t i = 0; i < n; ++i) {
				const double v = type_ == CV_64F ? ((const double *)data)[i] : (double)((const float *)data)[i];
				if (rtype == CV_64F) ((double *)m.data)[i] = v; else ((float *)m.data)[i] = (float)v;
			}
			dst = m;
		}
		int rows, cols;
		unsigned char *data;

	private:
		size_t elem() const { const int depth = type_ & 7, cn = (type_ >> 3) + 1; return (size_t)cn * (depth == CV_8U ? 1 : depth == CV_32F ? 4 : 8); }
		int type_;
		std::vector<unsigned char> store_;
};

}  // namespace cv
