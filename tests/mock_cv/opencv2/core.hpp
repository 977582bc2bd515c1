// Test double, NOT OpenCV: the handful of cv:: declarations include/vo_cv_adaptor.hpp touches, so that the adaptor can be
// compiled and run in a repository without OpenCV (tests/test_cpu_host.py).  Nothing of the reference is built with it.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_64F 6
#define CV_Assert(x) assert(x)
namespace cv {
struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float a, float b) : x(a), y(b) {}
};
struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0;
    DMatch() = default;
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};
class Mat {
   public:
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    size_t step = 0;
    Mat() = default;
    Mat(int r, int c, int type, void* p = nullptr) : rows(r), cols(c), type_(type) {
        step = (size_t)c * elem();
        if (p) {
            data = (uint8_t*)p;
        } else {
            own_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * step](), std::default_delete<uint8_t[]>());
            data = own_.get();
        }
    }
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + r * m.step, data + r * step, m.step);
        return m;
    }
    void convertTo(Mat& dst, int type) const {
        assert(type == type_);  // the mock only "converts" between equal types
        dst = clone();
    }
    template <class T>
    T& at(int r, int c) const { return *(T*)(data + r * step + c * sizeof(T)); }
    template <class T>
    T& at(int i) const { return *(T*)(data + (size_t)(i / cols) * step + (size_t)(i % cols) * sizeof(T)); }

   private:
    size_t elem() const { return type_ == CV_64F ? 8 : (type_ == CV_8UC3 ? 3 : 1); }
    int type_ = 0;
    std::shared_ptr<uint8_t> own_;
};
}  // namespace cv
