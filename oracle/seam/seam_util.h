// TEST INFRASTRUCTURE (oracle/seam) — the four declarations src/matrix.cc needs from src/util.h, whose real header drags
// in protobuf, MPI and CImg (compiled with -DUTIL_H_ -include seam_util.h, the technique of SURVEY.md §8b/§8c).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include <hdf5.h>
std::string GetStringError(int err_code);
void WriteHDF5CPU(hid_t file, float* mat, int rows, int cols, const std::string& name);
void ReadHDF5CPU(hid_t file, float* mat, int size, const std::string& name);
void ReadHDF5Shape(hid_t file, const std::string& name, int* rows, int* cols);
