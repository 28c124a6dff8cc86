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

    def GetBatch(self, data_layers):
        b = self.batches_[self.pos_ % len(self.batches_)]
        self.pos_ += 1
        for l in data_layers:
            (l.GetState() if l.IsInput() else l.GetData()).Set(b[l.GetName()])
