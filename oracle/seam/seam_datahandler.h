// TEST INFRASTRUCTURE (oracle/seam) — stands in for the reference's src/datahandler.h + src/datawriter.h (whose real versions
// pull in OpenCV / CImg / JPEG / the disk preloader) so that the reference's src/convnet.cc and src/grad_check.cc compile
// UNMODIFIED: compiled with -DDATAHANDLER_H_ -DDATAWRITER_H_ -include seam_datahandler.h.  Same class names and the methods
// convnet.cc calls (convnet.cc:463-536,571-657); the data itself is synthetic and deterministic: inputs unit-variance uniform from a counter
// hash, labels uniform in [0, classes), `dataset_size` cases generated batch by batch from (seed, batch index) so
// the CPU-linked and the GPU-linked builds see identical batches.
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "layer.h"

class DataHandler {
 public:
  explicit DataHandler(const config::DatasetConfig& config)
      : batch_size_(config.batch_size()), dataset_size_(config.max_dataset_size() > 0 ? config.max_dataset_size() : 4 * config.batch_size()),
        multiplicity_(config.multiplicity()), pos_(0), seed_(config.chunk_size() > 0 ? config.chunk_size() : 1) {}
  virtual ~DataHandler() {
    for (auto& kv : cache_) delete kv.second;
  }

  void GetBatch(std::vector<Layer*>& data_layers) {
    const int batch_index = pos_ / batch_size_;
    for (Layer* l : data_layers) {
      Matrix& dest = l->IsInput() ? l->GetState() : l->GetData();
      // every distinct batch is generated once and kept as a Matrix (device-resident in the GPU build), like the reference's
      // own handler keeps its current chunk on the device (src/datahandler.cc:145-198): a step then costs one device copy
      Matrix*& cached = cache_[std::make_pair(batch_index, l->GetName())];
      if (cached == NULL) {
        cached = new Matrix();
        cached->AllocateGPUMemory(dest.GetRows(), dest.GetCols());
        Fill(cached->GetHostData(), (size_t)dest.GetRows() * dest.GetCols(), batch_index, l->IsInput(), l->GetNumChannels());
        cached->CopyToDevice();
      }
      dest.Set(*cached);
    }
    pos_ += batch_size_;
    if (pos_ + batch_size_ > dataset_size_) pos_ = 0;
  }
  int GetBatchSize() const { return batch_size_; }
  int GetDataSetSize() const { return dataset_size_; }
  int GetMultiplicity() const { return multiplicity_; }
  void Seek(int row) { pos_ = row; }
  void Sync() {}
  void SetFOV(const int, const int, const int, const int, const int, const int, const int) {}
  void AllocateMemory() {}

 private:
  // counter-based generator (lowbias32 hash of the flat element index), restated in tests/ref_host.py so the python host can
  // be fed the very same batches: inputs uniform with zero mean and unit variance, labels h % classes
  void Fill(float* h, size_t n, int batch_index, bool is_input, int classes) const {
    const unsigned base = seed_ * 0x9E3779B1u + (unsigned)batch_index * 0x85EBCA77u + (is_input ? 0x1234567u : 0x7654321u);
    for (size_t i = 0; i < n; ++i) {
      unsigned x = base + (unsigned)i;
      x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
      if (is_input) h[i] = ((float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4641016f;
      else h[i] = (float)(x % (unsigned)classes);
    }
  }
  std::map<std::pair<int, std::string>, Matrix*> cache_;
  int batch_size_, dataset_size_, multiplicity_, pos_;
  unsigned seed_;
};

class DataWriter {
 public:
  explicit DataWriter(const config::FeatureExtractorConfig&) {}
  void SetDataSetSize(int) {}
  void SetNumDims(const std::string&, const int) {}
  void Write(std::vector<Layer*>&, int) {}
};
