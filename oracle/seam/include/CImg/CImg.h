/* TEST INFRASTRUCTURE (oracle/seam) — stands in for the CImg library header so the reference's src/util.h (and with it the
 * operator sources) compile: util.h only names these two types in declarations (util.h:60,77,86). */
#pragma once
namespace cimg_library {
template <typename T> class CImg {};
class CImgDisplay {};
}  // namespace cimg_library
