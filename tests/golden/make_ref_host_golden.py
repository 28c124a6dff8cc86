"""Generates tests/golden/ref_host_tiny_alex.npz by running the REFERENCE'S OWN HOST on the reference's CPU path
(oracle/_ref/libref_host_cpu.so: src/convnet.cc, layer.cc, every *_edge.cc, optimizer.cc, loss_functions.cc, CPUMatrix.cc, eigenmat —
all compiled unmodified; `make -C oracle host`, needs /root/reference) on the AlexNet-topology test net of tests/test_net_gpu.py:

    p0      the reference's own initialisation (flat parameter buffer, 128-float aligned slices, src/convnet.cc:271-296)
    g0      flat gradient after one Fprop(train)/ComputeDeriv/Bprop on batch 0 at p0
    p3      parameters after 3 x ConvNet::TrainOneBatch from p0 (momentum SGD with l2 decay, batches 0,1,0)
    loss3   the loss layer's value after each of those steps;  correct3 = summed correct count

`ref_host_dag.npz` is the same for the two-branch merging net `dag_net()` of tests/test_reference_host.py.

Run from the repo root:  python tests/golden/make_ref_host_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_host  # noqa: E402
from test_net_gpu import small_alexnet  # noqa: E402

BATCH, NUM_BATCHES, SEED, STEPS = 8, 2, 5, 3

def make(host, text, out_name):
    with tempfile.TemporaryDirectory() as tmp:
        m, d = ref_host.write_configs(tmp, text, BATCH, NUM_BATCHES, SEED)
        p0 = host.init_params(m, d)
        g0 = host.gradient(m, d, p0)
        p3, correct, loss = host.train(m, d, STEPS, p0)
    out = os.path.join(HERE, out_name)
    np.savez_compressed(out, p0=p0, g0=g0, p3=p3, loss3=loss, correct3=np.float32(correct),
                        cfg=np.array([BATCH, NUM_BATCHES, SEED, STEPS], np.int32))
    print(out, p0.size, "params", os.path.getsize(out), "bytes", "loss", loss)


if __name__ == "__main__":
    from test_reference_host import dag_net
    host = ref_host.RefHost(ref_host.CPU_SO)
    make(host, small_alexnet(), "ref_host_tiny_alex.npz")
    # the merging DAG of tests/test_reference_host.py::dag_net (a layer with two incoming edges): same recipe
    make(host, dag_net(), "ref_host_dag.npz")
