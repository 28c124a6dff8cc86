"""ConvNet — mirror of src/convnet.{h,cc} for the training hot path: build the graph from a pbtxt
model, topologically sort, allocate the flat parameter / gradient buffers, Fprop / Bprop /
UpdateWeights / TrainOneBatch.

What changed relative to the reference, and why (MI355X-first):
  * Gradient exchange.  The reference D2H-copies the whole 249 MB gradient, sums it on rank 0 through
    MPI_Send/Recv, divides, and MPI_Bcasts it back (src/convnet.cc:407-450) *after* Bprop returns.
    Here every edge's gradient slice is all-reduced by RCCL (torch.distributed "nccl" backend over
    xGMI) on a communication stream the moment ComputeOuter has produced it, overlapping the rest
    of the backward pass; UpdateWeights waits per slice right before its optimizer step
    (data_parallel.GradientExchange).
  * GetLoss no longer forces a device->host sync every step (src/convnet.cc:482 -> matrix.cc:253-268):
    with ``fused=True`` the correct-count accumulates on device and is read every ``print_after``.
  * ``fused=True`` routes conv+bias+ReLU, FC+bias+ReLU, ReLU+dropout, softmax+CE-deriv+count and
    the SGD step through the library's fused entry points.  ``fused=False`` issues exactly the
    reference's Matrix-call sequence (used by the parity tests and grad_check).
"""
import sys
from collections import deque

import torch

from . import pbtxt
from .edge import ConvEdge, Edge, EdgeWithWeight, FCEdge, ResponseNormEdge
from .layer import Layer, SoftmaxLayer
from .matrix import Matrix
from .trainer import TrainLoopMixin


class ConvNet(TrainLoopMixin):
    def __init__(self, model, fused=False, process_id=0, num_processes=1, verbose=False, exchange=None, overlap_update=None,
                 overlap_wgrad=None):
        """``model``: path to a pbtxt file, pbtxt text, or a parsed pbtxt.Model.
        ``overlap_update`` (default off): inside TrainOneBatch each edge's optimizer step is enqueued on a second
        HIP stream as soon as that edge's wgrad + dgrad (and, data-parallel, its gradient bucket's all-reduce)
        are done.  Same arithmetic as the reference's serial UpdateWeights (bit-identical, tested).  Measured on
        one MI355X it is throughput-neutral (18.34 vs 18.36 ms/step): the co-running HBM-bound update kernels take
        CU slots from the MFMA kernels and slow them by what they save, so it stays opt-in.
        ``overlap_wgrad`` (default off): inside TrainOneBatch every edge's ComputeOuter (weight gradient, matrix-pipe bound) is
        enqueued on the second stream, ordered after the derivative it reads; the main stream goes straight on to the edge's
        ComputeDown and the next layers' backward (pool / response-norm undo are HBM-bound and can share the chip with it).
        The optimizer step (on either stream) and the gradient exchange are ordered behind it.  Same arithmetic."""
        self.verbose = verbose
        self.fused = fused
        self.overlap_update_ = bool(overlap_update)
        self.overlap_wgrad_ = bool(overlap_wgrad)
        # Where a conv edge's weight gradient joins the second stream.  "before": as soon as the edge's output derivative is final, so it
        # runs beside the edge's own ComputeDown; "after": behind that ComputeDown, beside what follows it on the main stream (the previous
        # layer's response-norm / pool undo).  Same arithmetic; which pairing is faster is a property of the kernels of the day: round 3
        # measured "after" 0.08 ms ahead; with round 4's kernels (faster gather-GEMMs, the 2 x 2-block pool undo) "before" is 0.25 ms ahead
        # (10.98-11.00 vs 11.23-11.28 ms, same call, profiles/r04_kernel_experiments.md §4).  CONVNET_WGRAD_ORDER overrides (A/B runs).
        import os
        self.wgrad_order_ = os.environ.get("CONVNET_WGRAD_ORDER", "before")
        self.side_stream_ = None
        self._pending_updates = []     # [(edge, event on the main stream after its dgrad, held back?)]
        self._in_train_step = False
        self.process_id_ = process_id
        self.num_processes_ = num_processes
        self.is_root_ = process_id == 0
        self.exchange_ = exchange
        if isinstance(model, pbtxt.Model):
            self.model_ = model
        elif "\n" in model or "{" in model:
            self.model_ = pbtxt.parse(model)
        else:
            self.model_ = pbtxt.read(model)
        m = self.model_
        # default optimizers for weights/biases whose optimizer is not specified (src/convnet.cc:36-50)
        for e in m.edge:
            if Edge.HasParameters(e):
                w_opt = m.default_weight_optimizer.copy()
                w_opt.MergeFrom(e.weight_optimizer)
                e.mutable("weight_optimizer").CopyFrom(w_opt)
                if not e.has_no_bias:
                    b_opt = m.default_bias_optimizer.copy()
                    b_opt.MergeFrom(e.bias_optimizer)
                    e.mutable("bias_optimizer").CopyFrom(b_opt)
        Matrix.InitRandom(m.seed + process_id)   # src/convnet.cc:67
        self.model_name_ = m.name
        self.layers_, self.edges_ = [], []
        self.input_layers_, self.output_layers_, self.data_layers_ = [], [], []
        self.batch_size_ = 0
        self.current_iter_ = 0
        self.parameters_, self.grad_parameters_, self.history_ = Matrix(), Matrix(), Matrix()
        self.edge_slices_ = {}   # edge -> (offset, length) in the flat buffers
        self.train_dataset_ = None
        self.correct_accum_ = None
        self._logits_pending = set()   # output layers whose state still holds logits (fused softmax)
        self.BuildNet()

    def log(self, *a):
        if self.verbose:
            print(*a, file=sys.stderr)

    # ---- graph: src/convnet.cc:150-242 ---------------------------------------------------------------
    def BuildNet(self):
        m = self.model_
        self.layers_ = [Layer.ChooseLayerClass(l) for l in m.layer]
        self.edges_ = [Edge.ChooseEdgeClass(e) for e in m.edge]
        for e in self.edges_:
            e.fused = self.fused
            if isinstance(e, EdgeWithWeight):
                e.weight_optimizer_.fused = self.fused
                if e.bias_optimizer_ is not None:
                    e.bias_optimizer_.fused = self.fused
        by_name = {e.GetName(): e for e in self.edges_}
        for e in self.edges_:
            if e.IsTied():
                e.SetTiedTo(by_name[e.GetTiedEdgeName()])
        for l in self.layers_:
            for e in self.edges_:
                if l.GetName() == e.GetSourceName():
                    l.AddOutgoing(e)
                    e.SetSource(l)
                    e.SetInputChannels(l.GetNumChannels())
                if l.GetName() == e.GetDestName():
                    l.AddIncoming(e)
                    e.SetDest(l)
                    e.SetOutputChannels(l.GetNumChannels())
        self.Sort()
        for l in self.layers_:
            if not l.incoming_edge_:
                self.input_layers_.append(l)
                self.data_layers_.append(l)
            if not l.outgoing_edge_:
                self.output_layers_.append(l)
                self.data_layers_.append(l)
        for l in self.layers_:
            if l.IsInput():
                y, x, t = l.GetSizeY(), l.GetSizeX(), l.GetSizeT()
                if y <= 0:
                    y = m.patch_size
                if x <= 0:
                    x = m.patch_size
                if t <= 0:
                    t = 1
            else:
                e0 = l.incoming_edge_[0]
                y, x, t = e0.GetNumModulesY(), e0.GetNumModulesX(), e0.GetNumModulesT()
            l.SetSize(y, x, t)
            self.log(f"Layer {l.GetName()}: {y}x{x}")
            for e in l.outgoing_edge_:
                e.SetImageSize(y, x, t)

    def Sort(self):
        # breadth-first topological sort, src/convnet.cc:312-353
        L, S = [], deque(l for l in self.layers_ if l.IsInput())
        if not S:
            raise SystemExit("Error: No layer is set to be input!")
        while S:
            n = S.popleft()
            L.append(n)
            for e in n.outgoing_edge_:
                e.SetMark()
                mm = e.GetDest()
                if mm is None:
                    raise SystemExit(f"Edge {e.GetName()} has no destination layer")
                if all(f.HasMark() for f in mm.incoming_edge_):
                    S.append(mm)
        if not all(f.HasMark() for f in self.edges_):
            raise SystemExit("Error : Network has loop(s)!")
        self.layers_ = L

    def GetLayerByName(self, name):
        for l in self.layers_:
            if l.GetName() == name:
                return l
        raise SystemExit(f"Error: No layer called {name}")

    def GetEdgeByName(self, name):
        for e in self.edges_:
            if e.GetName() == name:
                return e
        raise SystemExit(f"Error: No edge called {name}")

    # ---- memory: src/convnet.cc:266-310 -----------------------------------------------------------------
    def SetBatchsize(self, batch_size):
        self.batch_size_ = batch_size

    def AllocateMemory(self, fprop_only=False):
        self.AllocateLayerMemory()
        self.AllocateEdgeMemory(fprop_only)
        if self.fused:
            self.correct_accum_ = Matrix()
            self.correct_accum_.AllocateGPUMemory(1, 1, "correct count")
            self.correct_accum_.Set(0.0)

    def AllocateLayerMemory(self):
        for l in self.layers_:
            l.AllocateMemory(self.batch_size_)

    def AllocateEdgeMemory(self, fprop_only):
        total, usage = 0, {}
        for e in self.edges_:
            mem = e.GetParameterMemoryRequirement()
            usage[e] = mem
            total += ((mem + 127) // 128) * 128   # 128-float aligned slices (src/convnet.cc:279)
        self.parameters_.AllocateGPUMemory(1, total, "parameters")
        if not fprop_only:
            self.grad_parameters_.AllocateGPUMemory(1, total, "grad parameters")
            self.grad_parameters_.Set(0.0)
            self.history_.AllocateGPUMemory(1, total, "optimizer history")   # flat, same layout
        offset = 0
        for e in self.edges_:
            mem = usage[e]
            if mem == 0:
                continue
            s = Matrix()
            self.parameters_.GetSlice(s, offset, offset + mem)
            e.SetMemory(s)
            if not fprop_only:
                g, h = Matrix(), Matrix()
                self.grad_parameters_.GetSlice(g, offset, offset + mem)
                self.history_.GetSlice(h, offset, offset + mem)
                e.SetGradMemory(g, h)
            self.edge_slices_[e] = (offset, mem)
            offset += ((mem + 127) // 128) * 128
        if self.is_root_ or self.exchange_ is None:
            for e in self.edges_:
                e.Initialize()
        if self.exchange_ is not None:
            self.exchange_.Broadcast(self.parameters_)   # src/convnet.cc:309
            if not fprop_only:
                self.exchange_.Register(self)

    def NumParameters(self):
        return sum(n for _, n in self.edge_slices_.values())

    # ---- fprop / bprop: src/convnet.cc:355-405 ---------------------------------------------------------
    def _can_fuse_up(self, l):
        if not self.fused or len(l.incoming_edge_) != 1 or isinstance(l, SoftmaxLayer):
            return False
        e = l.incoming_edge_[0]
        if isinstance(e, ConvEdge):
            return e.has_no_bias_ or e.shared_bias_
        if isinstance(e, ResponseNormEdge):
            return l.is_relu       # the ReLU of an rnorm-fed layer rides in the rnorm kernel; other activations do not
        return isinstance(e, FCEdge)

    def Fprop(self, train):
        for l in self.layers_:
            fused_act = self._can_fuse_up(l)
            for e in l.incoming_edge_:
                src = e.GetSource()
                overwrite = l.AddOrOverwriteState(e.GetDestSliceName())
                e.ComputeUp(src.GetState(), l.GetState(), overwrite, train, fuse_relu=(l.is_relu if fused_act else None))
            if not l.IsInput() and not fused_act:
                if self.fused and isinstance(l, SoftmaxLayer) and l.IsOutput() and train:
                    self._logits_pending.add(l)   # softmax + CE derivative + correct count are fused in ComputeDeriv
                else:
                    l.ApplyActivation()
            l.ApplyDropout(train)

    def _bprop_edge(self, output, input, edge, fuse_mask=None):
        # ConvNet::Bprop(output, input, edge), src/convnet.cc:362-375
        if edge.IsBackPropBlocked():
            return
        side = self._in_train_step and self.side_stream_ is not None
        if side and isinstance(edge, ConvEdge) and any(held for _, _, held in self._pending_updates):
            # the FC updates held back so far start now, beside this conv edge's MFMA-bound backward
            now = torch.cuda.Event()
            now.record(torch.cuda.current_stream())
            self._pending_updates = [(e, now, False) for e, _, _ in self._pending_updates]
            self._flush_updates(final=False)
        # The gradient slice belongs to the OWNER of the weights: an edge tied to another one (tied_to) accumulates into its
        # owner's slice (edge_with_weight.cc:66-90), so the slice is final only when the last sharing edge has added its part —
        # whichever of them comes last in backward order.
        owner = edge.tied_edge_ if edge.IsTied() else edge

        def dgrad():
            if input.IsInput():
                return
            overwrite = input.AddOrOverwriteDeriv(edge.GetSourceSliceName())
            if fuse_mask is not None:
                edge.ComputeDown(output.GetDeriv(), input.GetState(), output.GetState(), input.GetDeriv(), overwrite, fuse_mask=fuse_mask)
            else:
                edge.ComputeDown(output.GetDeriv(), input.GetState(), output.GetState(), input.GetDeriv(), overwrite)

        if side and self.overlap_wgrad_ and isinstance(edge, EdgeWithWeight):
            # weight gradient on the second stream, behind everything enqueued so far (the derivative it reads); the all-reduce of
            # a completed slice is posted from that stream too, so its `ready` event covers the wgrad.  ComputeOuter reads the
            # source layer's state and the destination's derivative; ComputeDown writes the SOURCE's derivative: either order of the
            # two is legal, and nothing later in this step writes what the weight gradient reads.
            late = self.wgrad_order_ == "after" and isinstance(edge, ConvEdge)
            if late:
                dgrad()
            here = torch.cuda.Event()
            here.record(torch.cuda.current_stream())
            self.side_stream_.wait_event(here)
            with Matrix.OnStream(self.side_stream_):
                edge.ComputeOuter(input.GetState(), output.GetDeriv())
                complete = isinstance(owner, EdgeWithWeight) and owner.GetNumGradsReceived() >= owner.num_shares_
                if self.exchange_ is not None and owner in self.edge_slices_ and complete:
                    self.exchange_.GradReady(owner)
            if not late:
                dgrad()
        else:
            edge.ComputeOuter(input.GetState(), output.GetDeriv())
            complete = isinstance(owner, EdgeWithWeight) and owner.GetNumGradsReceived() >= owner.num_shares_
            if self.exchange_ is not None and owner in self.edge_slices_ and complete:
                self.exchange_.GradReady(owner)     # the slice is final: start its all-reduce
            dgrad()
        if side and self.overlap_update_ and isinstance(owner, EdgeWithWeight) and complete:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())   # wgrad AND dgrad (which reads the weights) are enqueued
            # An FC edge's backward at batch <= a few hundred is itself HBM-bound (it streams the weight matrix
            # three times), so running its update beside it gains nothing: hold it until a conv edge's backward
            # (MFMA-bound) begins.  Conv updates are small and start at once.
            # (Weights shared by tied edges are also READ by the sharers' ComputeDown: every sharer's dgrad precedes the
            # completing ComputeOuter in program order except the completing edge's own, enqueued just above — so `ev` covers all.)
            self._pending_updates.append((owner, ev, isinstance(owner, FCEdge)))
            self._flush_updates(final=False)

    def _flush_updates(self, final):
        """Enqueue on the side stream the optimizer step of every pending edge whose gradient is final."""
        keep = []
        for edge, ev, held in self._pending_updates:
            if held and not final:
                keep.append((edge, ev, held))
                continue
            bucket_ev = None
            if self.exchange_ is not None and edge in self.edge_slices_:
                state = self.exchange_.BucketState(edge)
                if state == "pending":
                    if final:
                        raise RuntimeError(f"gradient bucket of {edge.GetName()} was never exchanged")
                    keep.append((edge, ev, held))
                    continue
                bucket_ev = state
            self.side_stream_.wait_event(ev)
            if bucket_ev is not None and not hasattr(bucket_ev, "wait_library_stream"):
                self.side_stream_.wait_event(bucket_ev)
            with Matrix.OnStream(self.side_stream_):
                if hasattr(bucket_ev, "wait_library_stream"):
                    bucket_ev.wait_library_stream()      # exchange through the library's own entries: it holds the done-events
                edge.UpdateWeights()
        self._pending_updates = keep

    def _fused_down_scale(self, l):
        """If layer l's dropout' + ReLU' can ride in its only outgoing edge's ComputeDown epilogue, return the
        scale to apply (else None).  The reference applies them after ALL outgoing edges have accumulated
        (src/convnet.cc:390-404), so this is only legal with exactly one edge."""
        if not self.fused or l.IsInput() or l.IsOutput() or not l.is_relu or len(l.outgoing_edge_) != 1:
            return None
        e = l.outgoing_edge_[0]
        if not e.can_fuse_mask or e.IsBackPropBlocked() or l.store_dropout_noise_:
            return None
        scale = 1.0 / (1 - l.dropprob_) if (l.dropprob_ > 0 and l.dropout_scale_up_at_train_time_) else 1.0
        from .edge import MaxPoolEdge
        if isinstance(e, MaxPoolEdge) and scale != 1.0:
            return None
        return scale

    def Bprop(self):
        for l in reversed(self.layers_):
            scale = self._fused_down_scale(l)
            for e in l.outgoing_edge_:
                self._bprop_edge(e.GetDest(), l, e, fuse_mask=scale)
            if scale is not None:
                continue   # dropout' and ReLU' were applied by the edge's epilogue
            l.ApplyDerivativeofDropout()
            if not l.IsInput() and not l.IsOutput():
                l.ApplyDerivativeOfActivation()

    def ComputeDeriv(self):
        for l in self.output_layers_:
            if l in self._logits_pending:
                self._logits_pending.discard(l)
                Matrix.SoftmaxCEGradCorrect(l.GetState(), l.GetData(), l.GetState(), l.GetDeriv(), self.correct_accum_,
                                            l.loss_function_weight_)
            else:
                l.ComputeDeriv()

    def GetLoss(self):
        """Per-output-layer performance metric (src/convnet.cc:456-461).  In fused mode a softmax output's correct count
        accumulates on device (no per-step sync; ReadCorrectCount) and GetLoss returns None — unless some output layer did not
        take the fused softmax path (e.g. a SQUARED_ERROR linear output) or there are several outputs: the on-device counter
        is one number, so those nets report per layer through the reference's own call, like the unfused path."""
        if self.fused and len(self.output_layers_) == 1 and isinstance(self.output_layers_[0], SoftmaxLayer):
            return None
        return [l.GetPerformanceMetric() for l in self.output_layers_]

    def TimestampModel(self):
        """ConvNet::TimestampModel (src/convnet.cc:830-838): stamp the run — checkpoints go to <dir>/<name>_<timestamp>.h5 — append
        the stamp to the model, write the stamped model as <dir>/<name>_<timestamp>.pbtxt and name the two log files."""
        import os
        import time
        from . import pbtxt
        ts = time.strftime("%Y%m%d%H%M%S")
        while ts in self.model_.timestamp:   # a resume within the same second must not reuse the name
            ts += "_"
        self.model_.timestamp.append(ts)
        fname = os.path.join(self.model_.checkpoint_dir, f"{self.model_.name}_{ts}")
        if self.model_.checkpoint_dir:
            os.makedirs(self.model_.checkpoint_dir, exist_ok=True)
            pbtxt.write(fname + ".pbtxt", self.model_)
        self.log_file_ = fname + "_train.log"
        self.val_log_file_ = fname + "_valid.log"
        return ts

    def ReadCorrectCount(self, reset=True):
        """Fused mode: number of correct predictions since the last read (one D2H sync)."""
        v = float(self.correct_accum_.ToNumpy().reshape(-1)[0])
        if reset:
            self.correct_accum_.Set(0.0)
        return v

    # ---- update: src/convnet.cc:440-450 -------------------------------------------------------------------
    def UpdateWeights(self):
        if self._in_train_step and self.side_stream_ is not None:
            if self.overlap_update_:
                # every edge went through _bprop_edge: drain what is still waiting for its bucket, then make the
                # main stream (next Fprop reads the weights) wait for the side stream
                self._flush_updates(final=True)
            done = torch.cuda.Event()
            done.record(self.side_stream_)
            torch.cuda.current_stream().wait_event(done)   # weight gradients (and side-stream updates) are in
            if self.overlap_update_:
                return
        # fused host: the plain SGD steps of every edge (AlexNet: five convolution banks and eight biases) leave as ONE launch behind the loop
        batch = [] if self.fused else None
        for e in self.edges_:
            if e.IsBackPropBlocked():
                continue
            if self.exchange_ is not None and e in self.edge_slices_:
                self.exchange_.WaitFor(e)       # the averaged gradient slice has arrived
            if batch is not None and isinstance(e, EdgeWithWeight):
                e.UpdateWeights(batch)
            else:
                e.UpdateWeights()
        if batch:
            Matrix.SGDMomentumStepMulti(batch)

    # ---- data -------------------------------------------------------------------------------------------
    def SetupDataset(self, dataset):
        self.train_dataset_ = dataset
        self.batch_size_ = dataset.GetBatchSize()

    def GetBatch(self, dataset):
        dataset.GetBatch(self.data_layers_)

    def TrainOneBatch(self):
        # src/convnet.cc:475-485
        for l in self.layers_:
            l.ResetAddOrOverwrite()
        for e in self.edges_:
            e.NotifyStart()
        for l in self.layers_:
            l.NotifyStart()
        if self.exchange_ is not None:
            self.exchange_.StartStep()
        if (self.overlap_update_ or self.overlap_wgrad_) and self.side_stream_ is None and torch.cuda.is_available():
            self.side_stream_ = Matrix.SharedStream("side")
        self.GetBatch(self.train_dataset_)
        self.Fprop(True)
        self.ComputeDeriv()
        error = self.GetLoss()
        self._in_train_step = True
        try:
            self.Bprop()
            self.UpdateWeights()
        finally:
            self._in_train_step = False
        self.current_iter_ += 1
        return error

    # ---- checkpoint / resume: src/convnet.cc:659-684,737-751 ------------------------------------------------------------
    def ReduceLearningRate(self, factor):
        # src/convnet.cc:826-830
        for e in self.edges_:
            if isinstance(e, EdgeWithWeight):
                e.ReduceLearningRate(factor)

    def GetCheckpointFilename(self):
        # src/convnet.cc:651-657: <checkpoint_dir>/<name>_<timestamp>.h5
        import os
        m = self.model_
        ts = m.timestamp[-1] if m.timestamp else ""
        return os.path.join(m.checkpoint_dir, f"{m.name}_{ts}.h5")

    def Save(self, output_file=None):
        """HDF5 file in the reference's layout: every edge's weight / bias / gradient_history datasets and step attributes,
        plus ``__lr_reduce_counter__`` and ``__current_iter__``; written to ``<name>temp`` and renamed (convnet.cc:666-684)."""
        import os
        from . import hdf5io
        if output_file is None:
            # ConvNet::Save() (src/convnet.cc:659-667): the checkpoint, then — with Polyak averaging on — the AVERAGED weights
            # beside it as <file>polyak, and the current weights restored
            fname = self.GetCheckpointFilename()
            self.Save(fname)
            if self.model_.polyak_after > 0:
                self.LoadPolyakWeights()
                self.Save(fname + "polyak")
                self.LoadCurrentWeights()
            return
        tmp = output_file + "temp"
        with hdf5io.File(tmp, "w") as f:
            for e in self.edges_:
                e.SaveParameters(f)
            f.WriteHDF5IntAttr("__lr_reduce_counter__", getattr(self, "lr_reduce_counter_", 0))
            f.WriteHDF5IntAttr("__current_iter__", self.current_iter_)
        os.replace(tmp, output_file)

    def Load(self, input_file=None):
        """Weights always; optimizer history + step only where the optimizer is allocated (a training net), so the same file
        serves resume and fprop-only use (edge_with_weight.cc:42-58).  Learning-rate reductions are re-applied."""
        from . import hdf5io
        with hdf5io.File(input_file or self.GetCheckpointFilename()) as f:
            for e in self.edges_:
                e.LoadParameters(f)
            self.lr_reduce_counter_ = f.ReadHDF5IntAttr("__lr_reduce_counter__", getattr(self, "lr_reduce_counter_", 0))
            for _ in range(self.lr_reduce_counter_):
                self.ReduceLearningRate(self.model_.reduce_lr_factor)
            self.current_iter_ = f.ReadHDF5IntAttr("__current_iter__", self.current_iter_)
