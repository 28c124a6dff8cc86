"""Synthetic DataHandler.  The reference's input pipeline (src/datahandler.cc: HDF5/JPEG/video
iterators, OpenCV) is out of hot-path scope; only its contract matters: ``GetBatch(data_layers)``
leaves a batch in every input layer's ``state_`` (N, X*Y*C CHWN) and every output layer's ``data_``
(src/datahandler.cc:145-198).  Inputs are N(0,1) (the real pipeline feeds mean/std-normalised
pixels, :496-507), labels uniform ints stored as float (N,1); ``num_batches`` distinct batches are
pre-generated on the device and cycled, so no host work sits inside a timed step."""
import numpy as np

from .matrix import Matrix


class SyntheticDataHandler:
    def __init__(self, net, batch_size, seed=0, num_batches=2, num_classes=None):
        self.batch_size_ = batch_size
        self.batches_ = []
        self.pos_ = 0
        rng = np.random.default_rng(seed)
        for _ in range(num_batches):
            b = {}
            for l in net.data_layers_:
                if l.IsInput():
                    dims = l.GetSizeY() * l.GetSizeX() * l.GetSizeT() * l.GetNumChannels()
                    m = Matrix()
                    m.AllocateGPUMemory(batch_size, dims)
                    m.FromNumpy(rng.standard_normal((dims, batch_size), dtype=np.float32))
                else:
                    k = num_classes or l.GetNumChannels()
                    m = Matrix()
                    m.AllocateGPUMemory(batch_size, 1)
                    m.FromNumpy(rng.integers(0, k, batch_size).astype(np.float32))
                b[l.GetName()] = m
            self.batches_.append(b)

    def GetBatchSize(self):
        return self.batch_size_

    def GetDataSetSize(self):
        return self.batch_size_ * len(self.batches_)

    def Seek(self, row):
        self.pos_ = row // self.batch_size_

    def Sync(self):
        pass

    def GetBatch(self, data_layers):
        b = self.batches_[self.pos_ % len(self.batches_)]
        self.pos_ += 1
        for l in data_layers:
            (l.GetState() if l.IsInput() else l.GetData()).Set(b[l.GetName()])


class ChunkDataHandler:
    """GPU-resident dataset chunk + per-batch staging on the GPU: DataHandler::GetBatch (src/datahandler.cc:146-198) and
    DataIterator::SampleNoise / AddNoise / Preprocess (:496-570) for one image stream and one label stream.

    The chunk sits on the GPU as a (dims, cases) matrix (one case per column, [colour][row][col] contiguous), exactly
    as the reference keeps it.  Every GetBatch: sample per-case crop offsets and flip bits on the device, then ONE
    ``extract_patches`` (crop + flip + transpose) writes the CHWN batch straight into the input layer's state — or
    ``copy_transpose`` when there is no jitter; when the chunk is exhausted the columns are shuffled in place by a fresh
    permutation (``ShuffleColumns``), images and labels alike.  Mean/std normalisation is applied once when the chunk is
    loaded, like ``DataIterator::Preprocess``.  Disk/HDF5/JPEG loading is not part of this row: ``images`` / ``labels``
    are numpy arrays."""

    def __init__(self, images, labels, batch_size, image_size, crop_size=None, colors=3, translate=True, flip=True, mean=None,
                 std=None, randomize=True, seed=0):
        images = np.ascontiguousarray(images, np.float32)
        self.chunk_size_, dims = images.shape
        assert dims == colors * image_size * image_size and labels.shape[0] == self.chunk_size_
        assert batch_size <= self.chunk_size_
        self.batch_size_ = batch_size
        self.image_size_, self.gpu_image_size_ = image_size, crop_size or image_size
        self.colors_, self.translate_, self.flip_, self.randomize_ = colors, translate, flip, randomize
        self.rng_ = np.random.default_rng(seed)
        # Sequential (non-shuffled) use — validation, feature extraction — reads the dataset round-robin, so the batch that
        # straddles the end continues with the first cases (the reference's chunk loader wraps the same way): keep a copy
        # of the first batch_size cases behind the last one so that window is one contiguous slice.
        pad = 0 if randomize else batch_size
        if pad:
            images = np.concatenate([images, images[:pad]], axis=0)
            labels = np.concatenate([np.asarray(labels).reshape(-1), np.asarray(labels).reshape(-1)[:pad]])
        self.data_ = Matrix()
        self.data_.AllocateGPUMemory(dims, self.chunk_size_ + pad)
        self.data_.FromNumpy(images)                       # numpy (cases, dims) = column-major (dims, cases)
        if mean is not None:                               # DataIterator::Preprocess: m.AddColVec(mean_, -1); m.DivideByColVec(std_)
            self.data_.AddColVec(self._colvec(mean, dims), -1)
            self.data_.DivideByColVec(self._colvec(std, dims))
        self.labels_ = Matrix()
        self.labels_.AllocateGPUMemory(1, self.chunk_size_ + pad)
        self.labels_.FromNumpy(np.asarray(labels, np.float32).reshape(-1))
        self.perm_ = Matrix()
        self.perm_.AllocateGPUMemory(1, self.chunk_size_)
        self.perm_host_ = np.arange(self.chunk_size_, dtype=np.float32)
        self.start_ = 0
        self.width_offset_, self.height_offset_, self.flip_bit_ = Matrix(), Matrix(), Matrix()
        for m in (self.width_offset_, self.height_offset_, self.flip_bit_):
            m.AllocateGPUMemory(1, batch_size)
        self.slice_, self.label_slice_ = Matrix(), Matrix()
        self.multiplicity_counter_ = 0

    @staticmethod
    def _colvec(v, dims):
        m = Matrix()
        m.AllocateGPUMemory(dims, 1)
        m.FromNumpy(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (dims,)) if np.size(v) != dims else np.asarray(v, np.float32))
        return m

    def GetBatchSize(self):
        return self.batch_size_

    def GetDataSetSize(self):
        return self.chunk_size_

    def Seek(self, row):
        self.start_ = row

    def Sync(self):
        pass

    def _jitter(self):
        return self.image_size_ != self.gpu_image_size_ or self.flip_

    def ShuffleIndices(self):
        # DataHandler::ShuffleIndices (:139-144): random_shuffle of the SAME index array every time, then to the device
        self.rng_.shuffle(self.perm_host_)
        self.perm_.FromNumpy(self.perm_host_)

    def SampleNoise(self):
        # DataIterator::SampleNoise (:533-570)
        if not self._jitter():
            return
        max_off = self.image_size_ - self.gpu_image_size_
        if self.translate_:
            self.height_offset_.FillWithRand()
            self.width_offset_.FillWithRand()
            self.height_offset_.Mult(max_off + 1)          # rounded down by the kernel's int()
            self.width_offset_.Mult(max_off + 1)
        else:                                              # centre / corner patches by multiplicity id
            mid = self.multiplicity_counter_ % 5
            w, h = [(max_off // 2, max_off // 2), (0, 0), (max_off, 0), (max_off, max_off), (0, max_off)][mid]
            self.height_offset_.Set(h)
            self.width_offset_.Set(w)
        if self.flip_:
            self.flip_bit_.FillWithRand()                  # flip if > 0.5
        else:
            self.flip_bit_.Set(self.multiplicity_counter_ // 5)

    def GetBatch(self, data_layers):
        end = self.start_ + self.batch_size_
        if end > self.chunk_size_ and self.randomize_:     # DataHandler::GetBatch :147-166: drop the tail, reshuffle, restart
            self.ShuffleIndices()
            self.data_.ShuffleColumns(self.perm_)
            self.labels_.ShuffleColumns(self.perm_)
            self.start_, end = 0, self.batch_size_
        self.SampleNoise()
        for l in data_layers:
            if l.IsInput():
                self.data_.GetSlice(self.slice_, self.start_, end)
                dest = l.GetState()
                if self._jitter():                         # DataIterator::AddNoise (:520-531)
                    Matrix.ExtractPatches(self.slice_, dest, self.width_offset_, self.height_offset_, self.flip_bit_, self.image_size_,
                                          self.image_size_, self.gpu_image_size_, self.gpu_image_size_)
                else:
                    self.slice_.CopyTranspose(dest)
            else:
                self.labels_.GetSlice(self.label_slice_, self.start_, end)
                self.label_slice_.CopyTranspose(l.GetData())
        self.start_ = end % self.chunk_size_ if not self.randomize_ else end
