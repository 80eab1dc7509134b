// Minimal stand-in for the two OpenCV value types in Dataloader's signatures
// (std::vector<cv::Point2f> getObservations(), cv::Mat getColor/getDepth;
// /root/reference/include/bundleadjust/Dataloader.h:16,22-23).  OpenCV is not installed here.
#pragma once
#include <vector>

namespace cv {
struct Point2f {
  float x, y;
  Point2f() : x(0), y(0) {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Mat {  // the BAL loader only ever returns an empty one (src/BalDataloader.cpp:103-109)
  bool empty() const { return true; }
};
}  // namespace cv
