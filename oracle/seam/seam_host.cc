// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// The reference's WHOLE HOST — src/convnet.cc, grad_check.cc, layer.cc, loss_functions.cc, optimizer.cc, edge.cc,
// edge_with_weight.cc and every *_edge.cc — compiled UNMODIFIED from where it lies under /root/reference and linked either to
//   (hip) the reference's GPU `class Matrix` (src/matrix.cc, unmodified) over THIS repo's libconvnet_hip.so, or
//   (cpu) the reference's CPU `class Matrix` (src/CPUMatrix.cc + eigenmat) — the oracle.
// So `ConvNet::Train` / `GradChecker::Run` of the reference run on the MI355X with no source change, and the same driver on
// the reference's CPU path gives the numbers to compare with.  Stand-ins (oracle/seam/): a generated config header instead of
// protoc's (gen_config_pb.py, from the reference's .proto), <cublas.h>, <CImg/CImg.h>, <google/protobuf/text_format.h>, and
// datahandler.h / datawriter.h replaced by a synthetic in-memory DataHandler (seam_datahandler.h).  This file supplies what
// src/util.cc would have (util.cc needs protobuf + CImg): the pbtxt reader over the generated classes, the HDF5 helpers (real,
// over libhdf5, as util.cc:128-208), and no-op display hooks; plus extern "C" trampolines for tests/test_reference_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <iostream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "convnet.h"
#include "edge_with_weight.h"
#include "grad_check.h"
#ifdef USE_CUDA
#include "seam_pools.h"
#endif

using std::string;
using std::vector;

// ---- src/util.h functions ------------------------------------------------------------------------------------------
template <class T>
void ReadPbtxt(const string& pbtxt_file, T& model) {
  std::ifstream f(pbtxt_file.c_str());
  if (!f) {
    std::cerr << "Could not open " << pbtxt_file << std::endl;
    exit(1);
  }
  std::stringstream ss;
  ss << f.rdbuf();
  model.ParseFromText(ss.str());
}
template <class T>
void WritePbtxt(const string&, const T&) {}
template void ReadPbtxt<config::Model>(const string&, config::Model&);
template void ReadPbtxt<config::DatasetConfig>(const string&, config::DatasetConfig&);
template void ReadPbtxt<config::FeatureExtractorConfig>(const string&, config::FeatureExtractorConfig&);
template void WritePbtxt<config::Model>(const string&, const config::Model&);

string GetStringError(int err_code) {
  char buf[64];
  snprintf(buf, sizeof buf, "cudamat error %d", err_code);
  return string(buf);
}

#ifndef USE_CUDA
// The reference's GPU Matrix allocates host memory with calloc (src/matrix.cc:146) and its optimizers rely on that: the momentum
// history is never cleared (src/optimizer.cc:135).  The CPU Matrix uses `new float[]` (src/CPUMatrix.cc:120), i.e. reads
// uninitialised memory on the first update.  Give the CPU build the same zeroed allocations (library-local: -Bsymbolic).
void* operator new[](size_t n) {
  void* p = calloc(n ? n : 1, 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete[](void* p) noexcept { free(p); }
#endif

#ifdef USE_CUDA
// src/matrix.cc:1152 asks the CUDA runtime which device is current; on this platform that is the HIP runtime
extern "C" int hipGetDevice(int* dev);
extern "C" cudaError_t cudaGetDevice(int* dev) { return hipGetDevice(dev); }
#endif

void AddVectors(vector<float>& a, vector<float>& b) {
  if (a.size() == 0) a.resize(b.size(), 0.f);
  for (size_t i = 0; i < a.size() && i < b.size(); ++i) a[i] += b[i];
}
string GetTimeStamp() { return "seam"; }
bool ReadLines(const string& filename, vector<string>& lines) {
  std::ifstream f(filename.c_str());
  if (!f) return false;
  string line;
  while (std::getline(f, line)) lines.push_back(line);
  return true;
}

void WriteHDF5CPU(hid_t file, float* mat, int rows, int cols, const string& name) {
  hsize_t dimsf[2] = {(hsize_t)rows, (hsize_t)cols};
  hid_t space = H5Screate_simple(2, dimsf, NULL);
  hid_t ds = H5Dcreate2(file, name.c_str(), H5T_NATIVE_FLOAT, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, mat);
  H5Sclose(space);
  H5Dclose(ds);
}
void WriteHDF5CPU(hid_t file, vector<float>& mat, int rows, int cols, const string& name) {
  WriteHDF5CPU(file, mat.data(), rows, cols, name);
}
void ReadHDF5Shape(hid_t file, const string& name, int* rows, int* cols) {
  hid_t ds = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
  hid_t space = H5Dget_space(ds);
  const int nd = H5Sget_simple_extent_ndims(space);
  hsize_t dims[2] = {1, 1};
  H5Sget_simple_extent_dims(space, dims, NULL);
  *cols = (int)dims[0];
  *rows = nd == 1 ? 1 : (int)dims[1];
  H5Sclose(space);
  H5Dclose(ds);
}
void ReadHDF5CPU(hid_t file, float* mat, int size, const string& name) {
  int rows, cols;
  ReadHDF5Shape(file, name, &rows, &cols);
  if (rows * cols != size) {
    std::cerr << "Dimension mismatch: Expected " << size << " Got " << rows << "-" << cols << std::endl;
    exit(1);
  }
  hid_t ds = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
  H5Dread(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, mat);
  H5Dclose(ds);
}
void WriteHDF5IntAttr(hid_t file, const string& name, const int* val) {
  hid_t aid = H5Screate(H5S_SCALAR);
  hid_t attr = H5Acreate2(file, name.c_str(), H5T_NATIVE_INT, aid, H5P_DEFAULT, H5P_DEFAULT);
  H5Awrite(attr, H5T_NATIVE_INT, val);
  H5Sclose(aid);
  H5Aclose(attr);
}
void ReadHDF5IntAttr(hid_t file, const string& name, int* val) {
  if (H5Aexists(file, name.c_str()) <= 0) return;
  hid_t attr = H5Aopen(file, name.c_str(), H5P_DEFAULT);
  H5Aread(attr, H5T_NATIVE_INT, val);
  H5Aclose(attr);
}

ImageDisplayer::ImageDisplayer() {}
ImageDisplayer::ImageDisplayer(int, int, int, bool, const string&) {}
void ImageDisplayer::DisplayImage(float*, int, int) {}
void ImageDisplayer::DisplayWeights(float*, int, int, int, bool) {}
void ImageDisplayer::DisplayLocalization(float*, float*, float*, int) {}
void ImageDisplayer::SetFOV(int, int, int, int, int, int, int) {}

// ---- trampolines ---------------------------------------------------------------------------------------------------------
namespace {
// access to the pieces of ConvNet the tests read back (all protected members, so: a subclass)
class SeamNet : public ConvNet {
 public:
  explicit SeamNet(const string& model_file) : ConvNet(model_file) {}
  Matrix& Params() { return parameters_; }
  Matrix& Grads() { return grad_parameters_; }
  vector<Layer*>& Layers() { return layers_; }
  vector<Edge*>& Edges() { return edges_; }
  bool ReduceLr(const vector<float>& v) { return CheckReduceLearningRate(v); }
  void OneStep(vector<float>& err) { TrainOneBatch(err); }
  float Loss() {   // what GradChecker::GetLoss reads (grad_check.cc:13-16), after the step's own Fprop
    float s = 0.f;
    for (Layer* l : output_layers_) s += l->GetLoss();
    return s;
  }
  void ForwardBackward() {
    for (Layer* l : layers_) l->ResetAddOrOverwrite();
    GetBatch(*train_dataset_);
    Fprop(true);
    ComputeDeriv();
    Bprop();
  }
  void NextBatch() { GetBatch(*train_dataset_); }
};

void setup_device() {
  static bool done = false;
  if (done) return;
  done = true;
#ifdef USE_CUDA
  Matrix::SetupCUDADevice(0);
  SeamPools::ReleaseAtExit();
#endif
}
}  // namespace

extern "C" {

// Builds the reference's ConvNet from `model_pbtxt` + `data_pbtxt`, copies `params_in` (if non-null; flat parameter buffer) in,
// runs `steps` x TrainOneBatch, and returns the parameter buffer, the summed per-step train metric (correct count for softmax),
// the per-step loss (loss_out[steps], read after each step's update) and the number of parameters.
// With steps == 0 it runs one Fprop/ComputeDeriv/Bprop instead and returns the flat GRADIENT buffer in params_out.
long seam_host_train(const char* model_pbtxt, const char* data_pbtxt, int steps, const float* params_in, float* params_out,
                     long params_cap, float* metric_out, float* loss_out) {
  setup_device();
  SeamNet net(model_pbtxt);
  net.SetupDataset(data_pbtxt);
  net.AllocateMemory(false);
  Matrix& P = net.Params();
  const long n = (long)P.GetRows() * P.GetCols();
  if (params_in) {
    memcpy(P.GetHostData(), params_in, sizeof(float) * n);
    P.CopyToDevice();
  }
  float metric = 0.f;
  if (steps < 0) {   // just report the size and the reference's own initialisation
    P.CopyToHost();
    if (params_out && n <= params_cap) memcpy(params_out, P.GetHostData(), sizeof(float) * n);
  } else if (steps == 0) {
    net.ForwardBackward();
    Matrix& G = net.Grads();
    G.CopyToHost();
    if (params_out && n <= params_cap) memcpy(params_out, G.GetHostData(), sizeof(float) * n);
  } else {
    vector<float> err;
    for (int i = 0; i < steps; ++i) {
      net.OneStep(err);
      for (float e : err) metric += e;
      if (loss_out) loss_out[i] = net.Loss();
    }
    P.CopyToHost();
    if (params_out && n <= params_cap) memcpy(params_out, P.GetHostData(), sizeof(float) * n);
  }
  if (metric_out) *metric_out = metric;
  return n;
}

// Wall-clock of the reference's own training loop body (ConvNet::TrainOneBatch, src/convnet.cc:475-485, including its per-step
// GetLoss read-back) on whichever Matrix this build links: `warmup` untimed steps, then `steps` timed between device syncs.
// Returns milliseconds per step; the last step's loss goes to loss_out[0].
double seam_host_bench(const char* model_pbtxt, const char* data_pbtxt, int warmup, int steps, float* loss_out) {
  setup_device();
  SeamNet net(model_pbtxt);
  net.SetupDataset(data_pbtxt);
  net.AllocateMemory(false);
  vector<float> err;
  for (int i = 0; i < warmup; ++i) net.OneStep(err);
  Matrix::SyncAllDevices();
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < steps; ++i) net.OneStep(err);
  Matrix::SyncAllDevices();
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (loss_out) loss_out[0] = net.Loss();
  return ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (steps > 0 ? steps : 1);
}

// Batch number `index` as the data handler shim hands it to the net: input state (N x dims, column-major) and labels.
long seam_host_batch(const char* model_pbtxt, const char* data_pbtxt, int index, float* x_out, long x_cap, float* y_out, long y_cap) {
  setup_device();
  SeamNet net(model_pbtxt);
  net.SetupDataset(data_pbtxt);
  net.AllocateMemory(false);
  for (int i = 0; i <= index; ++i) net.NextBatch();
  long nx = 0;
  for (Layer* l : net.Layers()) {
    if (l->IsInput()) {
      Matrix& m = l->GetState();
      m.CopyToHost();
      nx = (long)m.GetRows() * m.GetCols();
      if (x_out && nx <= x_cap) memcpy(x_out, m.GetHostData(), sizeof(float) * nx);
    } else if (l->IsOutput()) {
      Matrix& m = l->GetData();
      m.CopyToHost();
      const long ny = (long)m.GetRows() * m.GetCols();
      if (y_out && ny <= y_cap) memcpy(y_out, m.GetHostData(), sizeof(float) * ny);
    }
  }
  return nx;
}

// `steps` x TrainOneBatch from params_in, then ConvNet::Save(path) (src/convnet.cc:669-684): the reference's own checkpoint
// writer.  params_out receives the flat parameter buffer that was saved.
long seam_host_checkpoint(const char* model_pbtxt, const char* data_pbtxt, int steps, const float* params_in, const char* path,
                          float* params_out, long params_cap) {
  setup_device();
  SeamNet net(model_pbtxt);
  net.SetupDataset(data_pbtxt);
  net.AllocateMemory(false);
  Matrix& P = net.Params();
  const long n = (long)P.GetRows() * P.GetCols();
  if (params_in) {
    memcpy(P.GetHostData(), params_in, sizeof(float) * n);
    P.CopyToDevice();
  }
  vector<float> err;
  for (int i = 0; i < steps; ++i) net.OneStep(err);
  net.Save(path);
  P.CopyToHost();
  if (params_out && n <= params_cap) memcpy(params_out, P.GetHostData(), sizeof(float) * n);
  return n;
}

// ConvNet::CheckReduceLearningRate (src/convnet.cc:788-818) on every prefix of a validation-error history: out[i] = decision
// after i+1 validations, with the model's reduce_lr_num_steps / reduce_lr_threshold / smaller_is_better.
void seam_host_reduce_lr(const char* model_pbtxt, const float* errors, int n, int* out) {
  setup_device();
  SeamNet net(model_pbtxt);
  vector<float> hist;
  for (int i = 0; i < n; ++i) {
    hist.push_back(errors[i]);
    out[i] = net.ReduceLr(hist) ? 1 : 0;
  }
}

// What the reference's BuildNet / Sort / size inference / AllocateEdgeMemory make of a model: one line per layer in
// topological (Fprop) order, one line per edge in edge order with its parameter-memory requirement, then the flat buffer size.
long seam_host_describe(const char* model_pbtxt, const char* data_pbtxt, char* out, long cap) {
  setup_device();
  SeamNet net(model_pbtxt);
  net.SetupDataset(data_pbtxt);
  net.AllocateMemory(false);
  std::ostringstream ss;
  for (Layer* l : net.Layers())
    ss << "layer " << l->GetName() << " " << l->GetSizeY() << " " << l->GetSizeX() << " " << l->GetNumChannels() << " " << (l->IsInput() ? 1 : 0)
       << " " << (l->IsOutput() ? 1 : 0) << "\n";
  for (Edge* e : net.Edges())
    ss << "edge " << e->GetSource()->GetName() << " " << e->GetDest()->GetName() << " " << e->GetParameterMemoryRequirement() << "\n";
  ss << "params " << (long)net.Params().GetRows() * net.Params().GetCols() << "\n";
  const string text = ss.str();
  if ((long)text.size() + 1 <= cap) memcpy(out, text.c_str(), text.size() + 1);
  return (long)text.size();
}

// The reference's optimizer alone: Optimizer::ChooseOptimizer on a text-format config::Optimizer, then `steps` rounds of what
// EdgeWithWeight does per iteration — NotifyStart (edge_with_weight.cc:108-110) and Optimize (edge_with_weight.cc:96-106) — on
// one (rows x cols) parameter with the caller's per-step gradients.  params_out receives the parameter after every step.
void seam_host_sgd(const char* optimizer_text, int rows, int cols, int steps, const float* params_in, const float* grads, float* params_out) {
  setup_device();
  config::Optimizer cfg;
  cfg.ParseFromText(optimizer_text);
  Optimizer* opt = Optimizer::ChooseOptimizer(cfg);
  opt->AllocateMemory(rows, cols);
  Matrix w, g;
  w.AllocateGPUMemory(rows, cols);
  g.AllocateGPUMemory(rows, cols);
  const size_t n = (size_t)rows * cols;
  memcpy(w.GetHostData(), params_in, sizeof(float) * n);
  w.CopyToDevice();
  for (int t = 0; t < steps; ++t) {
    opt->NotifyStart(w);
    memcpy(g.GetHostData(), grads + (size_t)t * n, sizeof(float) * n);
    g.CopyToDevice();
    opt->Optimize(g, w);
    w.CopyToHost();
    memcpy(params_out + (size_t)t * n, w.GetHostData(), sizeof(float) * n);
  }
  delete opt;
}

// The reference's run_grad_check (apps/run_grad_check.cc): GradChecker on the model with its own grad_check flags.
void seam_host_grad_check(const char* model_pbtxt, int batch_size, const char* output_h5) {
  setup_device();
  GradChecker gc(model_pbtxt);
  gc.SetBatchsize(batch_size);
  gc.AllocateMemory(false);
  gc.Run(output_h5);
}

#ifdef USE_CUDA
}  // extern "C"  (closed around the class below)

// ---- data-parallel host on the library's exchange entries (include/convnet_hip.h, csrc/comm.hip) --------------------------------
// What a maintainer of the reference adds to train_convnet_data_parallel, shown here as a SUBCLASS so that src/convnet.cc stays
// untouched (INTEGRATION.md §4 prints the same thing as a patch): ConvNet::Bprop(output, input, edge) and ConvNet::UpdateWeights
// are virtual (src/convnet.h:125,136).  Every edge's slice of the flat gradient is posted for all-reduce the moment its
// ComputeOuter has run (slices coalesced into buckets of >= bucket_bytes), and UpdateWeights waits — on the device — for a slice's
// bucket right before that edge's optimizer step, instead of Accumulate + Broadcast of the whole buffer through the host
// (src/convnet.cc:407-450).  The prototypes are repeated because the reference's cudamat.cuh and this repo's convnet_hip.h both
// define struct cudamat (identically).
extern "C" {
int convnet_hip_comm_unique_id(char* id_out);
int convnet_hip_comm_init(int rank, int nranks, const char* id_in);
int convnet_hip_comm_broadcast(cudamat* mat, int root);
int convnet_hip_comm_allreduce_avg(cudamat* flat, size_t offset, size_t count, int slot);
int convnet_hip_comm_wait(int slot);
int convnet_hip_comm_sync(void);
int convnet_hip_comm_destroy(void);
}

namespace {
// Matrix::GetMat() is protected (src/matrix.h:217).  The real integration adds three public Matrix methods that forward to the
// exchange entries; this shim must leave src/matrix.h untouched, so it reaches the cudamat through a derived
// accessor instead.
struct MatAccess : public Matrix {
  static cudamat* Of(Matrix& m) { return static_cast<MatAccess&>(m).GetMat(); }
};

class SeamDPNet : public SeamNet {
 public:
  SeamDPNet(const string& model_file, size_t bucket_bytes) : SeamNet(model_file), bucket_floats_(bucket_bytes / sizeof(float)) {}
  void BroadcastParameters() { CheckRc(convnet_hip_comm_broadcast(MatAccess::Of(parameters_), 0)); }   // src/convnet.cc:309
  int Buckets() const { return next_slot_; }

 protected:
  void Bprop(Layer& output, Layer& input, Edge& edge) override {
    ConvNet::Bprop(output, input, edge);
    EdgeWithWeight* e = dynamic_cast<EdgeWithWeight*>(&edge);
    if (e == NULL || edge.IsBackPropBlocked()) return;
    if (edge.IsTied()) {
      std::cerr << "SeamDPNet: tied edges are not handled by this demonstration host" << std::endl;
      exit(1);
    }
    const size_t off = MatAccess::Of(e->GetGradWeight())->data_device - MatAccess::Of(grad_parameters_)->data_device;
    const size_t n = (edge.GetParameterMemoryRequirement() + 127) / 128 * 128;   // the slice with its alignment pad (convnet.cc:279)
    if (open_ && off + n != lo_) Flush();          // not adjacent (a DAG): close the bucket
    if (!open_) { hi_ = off + n; open_ = true; }
    lo_ = off;                                     // backward order walks the flat buffer downwards
    pending_.push_back(e);
    if (hi_ - lo_ >= bucket_floats_) Flush();
  }
  void UpdateWeights() override {
    Flush();
    for (Edge* ed : edges_) {
      if (ed->IsBackPropBlocked()) continue;
      std::map<Edge*, int>::iterator it = slot_of_.find(ed);
      if (it != slot_of_.end()) CheckRc(convnet_hip_comm_wait(it->second));
      ed->UpdateWeights();
    }
    slot_of_.clear();
    next_slot_ = 0;
  }

 private:
  void Flush() {
    if (!open_) return;
    size_t hi = hi_;
    const size_t total = (size_t)grad_parameters_.GetRows() * grad_parameters_.GetCols();
    if (hi > total) hi = total;
    CheckRc(convnet_hip_comm_allreduce_avg(MatAccess::Of(grad_parameters_), lo_, hi - lo_, next_slot_));
    for (Edge* e : pending_) slot_of_[e] = next_slot_;
    pending_.clear();
    ++next_slot_;
    open_ = false;
  }
  static void CheckRc(int rc) {
    if (rc != 0) {
      std::cerr << "convnet_hip_comm error " << rc << std::endl;
      exit(1);
    }
  }
  size_t bucket_floats_, lo_ = 0, hi_ = 0;
  bool open_ = false;
  int next_slot_ = 0;
  vector<Edge*> pending_;
  std::map<Edge*, int> slot_of_;
};
}  // namespace

extern "C" {

// `steps` x TrainOneBatch of the data-parallel host above as rank `rank` of `nranks` (id: the 128-byte RCCL id from rank 0's
// seam_host_dp_unique_id; with nranks == 1 the id may be NULL and is created here).  Returns the parameter count; *buckets_out =
// buckets posted in the last step.
long seam_host_train_dp(const char* model_pbtxt, const char* data_pbtxt, int steps, const float* params_in, float* params_out, long params_cap,
                        int rank, int nranks, const char* id, long bucket_bytes, int* buckets_out) {
  setup_device();
  char local_id[128];
  if (id == NULL) {
    if (nranks != 1 || convnet_hip_comm_unique_id(local_id) != 0) return -1;
    id = local_id;
  }
  if (convnet_hip_comm_init(rank, nranks, id) != 0) return -2;
  long n = 0;
  {
    SeamDPNet net(model_pbtxt, (size_t)bucket_bytes);
    net.SetupDataset(data_pbtxt);
    net.AllocateMemory(false);
    Matrix& P = net.Params();
    n = (long)P.GetRows() * P.GetCols();
    if (params_in) {
      memcpy(P.GetHostData(), params_in, sizeof(float) * n);
      P.CopyToDevice();
    }
    net.BroadcastParameters();
    vector<float> err;
    int buckets = 0;
    for (int i = 0; i < steps; ++i) {
      net.OneStep(err);
      buckets = net.Buckets();
    }
    (void)buckets;
    convnet_hip_comm_sync();
    P.CopyToHost();
    if (params_out && n <= params_cap) memcpy(params_out, P.GetHostData(), sizeof(float) * n);
  }
  convnet_hip_comm_destroy();
  return n;
}

int seam_host_dp_unique_id(char* id_out) { return convnet_hip_comm_unique_id(id_out); }

// seam_host_bench with the data-parallel host above as the only rank of a world of one: every bucket is posted through
// convnet_hip_comm_allreduce_avg from Bprop and waited for in UpdateWeights (events, comm stream, slot bookkeeping all live; the
// all-reduce itself is the identity and is skipped inside the library).  Milliseconds per step, or a negative error code.
double seam_host_bench_dp(const char* model_pbtxt, const char* data_pbtxt, int warmup, int steps, long bucket_bytes, float* loss_out) {
  setup_device();
  char id[128];
  if (convnet_hip_comm_unique_id(id) != 0) return -1.0;
  if (convnet_hip_comm_init(0, 1, id) != 0) return -2.0;
  double ms = 0.0;
  {
    SeamDPNet net(model_pbtxt, (size_t)bucket_bytes);
    net.SetupDataset(data_pbtxt);
    net.AllocateMemory(false);
    net.BroadcastParameters();
    vector<float> err;
    for (int i = 0; i < warmup; ++i) net.OneStep(err);
    convnet_hip_comm_sync();
    Matrix::SyncAllDevices();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < steps; ++i) net.OneStep(err);
    convnet_hip_comm_sync();
    Matrix::SyncAllDevices();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (loss_out) loss_out[0] = net.Loss();
    ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (steps > 0 ? steps : 1);
  }
  convnet_hip_comm_destroy();
  return ms;
}
#endif  // USE_CUDA

// run_grad_check on a FIXED point: GradChecker::Run (src/grad_check.cc:77-140) with its random fill of the inputs and labels
// replaced by the data shim's batch 0 and the parameters supplied by the caller, so the reference's CPU build and its build on
// this library differentiate the SAME function at the SAME point and their verdicts can be compared check by check.  Everything
// that decides — ComputeNumericGrad, the pass rule (grad_check.cc:37-75, including its carry-over of diff_sum), the HDF5 datasets —
// is the reference's own compiled GradChecker::GradCheck.  passed_out[2*i], [2*i+1] = weights / bias verdict of the i-th edge
// with grad_check: true; returns the number of such edges.
namespace {
class SeamGradChecker : public GradChecker {
 public:
  explicit SeamGradChecker(const string& model_file) : GradChecker(model_file) {}
  int RunFixed(const float* params, const string& output_file, int* passed_out, int cap) {
    if (params) {
      memcpy(parameters_.GetHostData(), params, sizeof(float) * (size_t)parameters_.GetRows() * parameters_.GetCols());
      parameters_.CopyToDevice();
    }
    hid_t file = H5Fcreate(output_file.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    for (Layer* l : layers_) l->ResetAddOrOverwrite();
    GetBatch(*train_dataset_);
    Fprop(false);
    ComputeDeriv();
    Bprop();
    int n = 0;
    for (Edge* ed : edges_) {
      if (!ed->GradCheck()) continue;
      EdgeWithWeight* e = dynamic_cast<EdgeWithWeight*>(ed);
      if (e == NULL) continue;
      const string& name = e->GetName();
      vector<float> eps_values;
      ed->GradCheckEpsilon(eps_values);
      Matrix& gw = e->GetGradWeight();
      Matrix& gb = e->GetGradBias();
      gw.CopyToHost();
      gb.CopyToHost();
      int nw = ed->GradCheckNumParams(), nb = ed->GradCheckNumParams();
      if (nw > gw.GetNumEls()) nw = gw.GetNumEls();
      if (nb > gb.GetNumEls()) nb = gb.GetNumEls();
      // the analytical gradients must outlive the perturbation loop (GetHostData of the live gradient matrix does, as in Run)
      WriteHDF5CPU(file, gw.GetHostData(), nw, 1, name + "_weights_analytical");
      WriteHDF5CPU(file, gb.GetHostData(), nb, 1, name + "_bias_analytical");
      const bool pw = GradCheck(e->GetWeight(), eps_values, nw, gw.GetHostData(), name + "_weights_numerical", file);
      const bool pb = GradCheck(e->GetBias(), eps_values, nb, gb.GetHostData(), name + "_bias_numerical", file);
      if (2 * n + 1 < cap) {
        passed_out[2 * n] = pw;
        passed_out[2 * n + 1] = pb;
      }
      ++n;
    }
    H5Fclose(file);
    return n;
  }
};
}  // namespace

int seam_host_grad_check_fixed(const char* model_pbtxt, const char* data_pbtxt, const float* params_in, const char* output_h5,
                               int* passed_out, int cap) {
  setup_device();
  SeamGradChecker gc(model_pbtxt);
  gc.SetupDataset(data_pbtxt);
  gc.AllocateMemory(false);
  return gc.RunFixed(params_in, output_h5, passed_out, cap);
}

}  // extern "C"
