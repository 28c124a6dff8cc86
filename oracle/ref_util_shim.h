// TEST INFRASTRUCTURE. Stand-in for the reference's src/util.h when compiling src/CPUMatrix.cc
// unmodified (-DUTIL_H_ -include ref_util_shim.h): the real header drags in protobuf + CImg,
// neither of which exists in this image. Only the four symbols CPUMatrix.cc uses are declared
// (reference src/util.h:40-60); definitions are no-op stubs in ref_shim.cc.
#pragma once
#include <hdf5.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>
std::string GetStringError(int err_code);
void WriteHDF5CPU(hid_t file, float* mat, int rows, int cols, const std::string& name);
void ReadHDF5CPU(hid_t file, float* mat, int size, const std::string& name);
void ReadHDF5Shape(hid_t file, const std::string& name, int* rows, int* cols);
