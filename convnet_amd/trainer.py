"""The callers on either side of TrainOneBatch: ConvNet::Train's loop (validate / reduce-lr / Polyak /
checkpoint cadence), Validate, CheckReduceLearningRate, the Polyak parameter queue and ExtractFeatures — host logic over
the same fprop / train-step kernels.  Mirrors src/convnet.cc:571-657 (Validate, ExtractFeatures), :686-735 (Polyak),
:788-830 (CheckReduceLearningRate), :866-1011 (Train).  Display / localisation hooks are not part of this row.

Dataset protocol (DataHandler): GetBatch(data_layers), GetBatchSize(), GetDataSetSize(), Seek(i), Sync()."""
import sys
import time

import numpy as np


class TrainLoopMixin:
    # ---- validation: src/convnet.cc:571-589 -------------------------------------------------------------------------
    def Validate(self, dataset=None):
        """Running mean over the dataset's full batches of every output layer's performance metric per case
        (accuracy for the softmax/classification loss, layer.cc GetPerformanceMetric)."""
        dataset = dataset if dataset is not None else getattr(self, "val_dataset_", None)
        if dataset is None:
            return []
        dataset.Seek(0)
        batch_size = dataset.GetBatchSize()
        num_batches = dataset.GetDataSetSize() // batch_size
        total = []
        for k in range(num_batches):
            for l in self.layers_:
                l.ResetAddOrOverwrite()
            dataset.GetBatch(self.data_layers_)
            self.Fprop(False)
            error = [l.GetPerformanceMetric() for l in self.output_layers_]
            if len(total) != len(error):
                total = [0.0] * len(error)
            total = [(t * k) / (k + 1) + e / (batch_size * (k + 1)) for t, e in zip(total, error)]
        dataset.Sync()
        return total

    # ---- learning-rate schedule on the validation curve: src/convnet.cc:788-817 ---------------------------------------
    def CheckReduceLearningRate(self, val_error):
        num_steps = self.model_.reduce_lr_num_steps
        n = len(val_error)
        if n < num_steps:
            return False
        i = n - num_steps
        mean1 = mean2 = 0.0
        for j in range(num_steps // 2):
            mean1 = (mean1 * j) / (j + 1) + val_error[i] / (j + 1)
            i += 1
        for j in range(num_steps - num_steps // 2):
            mean2 = (mean2 * j) / (j + 1) + val_error[i] / (j + 1)
            i += 1
        diff = mean1 - mean2 if self.model_.smaller_is_better else mean2 - mean1
        return diff < self.model_.reduce_lr_threshold

    # ---- Polyak averaging: parameter snapshots live in host memory (src/convnet.cc:686-735) ---------------------------
    def _polyak_state(self):
        if not hasattr(self, "polyak_parameters_"):
            self.polyak_queue_size_ = self.model_.polyak_queue_size
            self.polyak_parameters_ = [None] * self.polyak_queue_size_
            self.polyak_index_, self.polyak_queue_full_, self.parameters_backup_ = 0, False, None
        return self

    def InsertPolyak(self):
        s = self._polyak_state()
        if s.polyak_queue_size_ == 0:
            return
        s.polyak_parameters_[s.polyak_index_] = self.parameters_.ToNumpy()
        s.polyak_index_ += 1
        if s.polyak_index_ == s.polyak_queue_size_:
            s.polyak_index_, s.polyak_queue_full_ = 0, True

    def LoadPolyakWeights(self):
        s = self._polyak_state()
        if s.polyak_queue_size_ == 0:
            return
        s.parameters_backup_ = self.parameters_.ToNumpy()
        max_ind = s.polyak_queue_size_ if s.polyak_queue_full_ else s.polyak_index_
        if max_ind == 0:
            return
        avg = np.zeros_like(s.parameters_backup_)
        for w in s.polyak_parameters_[:max_ind]:       # fp32 running sum in queue order, then one divide (:724-729)
            avg += w
        avg /= np.float32(max_ind)
        self.parameters_.FromNumpy(avg)

    def LoadCurrentWeights(self):
        if getattr(self, "parameters_backup_", None) is not None:
            self.parameters_.FromNumpy(self.parameters_backup_)

    # ---- the training loop: src/convnet.cc:866-1011 --------------------------------------------------------------------
    def SetupValidationDataset(self, dataset):
        self.val_dataset_ = dataset

    def Train(self, max_iter=None, log=None, checkpoint=True):
        """Runs from ``current_iter_`` to ``max_iter`` (default model.max_iter) with the reference's cadence: train accuracy
        every ``print_after`` steps, validation every ``validate_after`` (on Polyak-averaged weights when ``polyak_after``
        is set, and — as the reference does, :966 has the restore commented out — training continues from them), learning
        rate cut by ``reduce_lr_factor`` when the validation curve flattens, checkpoint every ``save_after``.
        Returns {"train": [(iter, acc…)], "val": [(iter, acc…)], "lr_reductions": n}."""
        if self.train_dataset_ is None:
            raise SystemExit("Error: Train dataset is NULL.")
        m = self.model_
        log = log or (lambda s: print(s, file=sys.stderr))
        max_iter = m.max_iter if max_iter is None else max_iter
        print_after, validate_after, save_after, polyak_after = m.print_after, m.validate_after, m.save_after, m.polyak_after
        start_polyak_queue_val = validate_after - polyak_after * m.polyak_queue_size
        start_polyak_queue_save = save_after - polyak_after * m.polyak_queue_size
        lr_reduce_layer_id = 0
        if m.reduce_lr_layer_name:
            names = [l.GetName() for l in self.output_layers_]
            if m.reduce_lr_layer_name not in names:
                raise SystemExit(f"No such output layer {m.reduce_lr_layer_name}")
            lr_reduce_layer_id = names.index(m.reduce_lr_layer_name)
        if not hasattr(self, "lr_reduce_counter_"):
            self.lr_reduce_counter_ = 0
        if self.is_root_ and checkpoint:
            self.TimestampModel()                 # src/convnet.cc:875: every Train() on root gets its own stamp — a resumed
                                                  # run checkpoints beside the file it was loaded from, never over it
        history = {"train": [], "val": [], "lr_reductions": 0}
        train_error, val_error = None, []
        dont_reduce_lr = 0
        start_t = time.time()
        val_dataset = getattr(self, "val_dataset_", None)
        if self.fused:
            self.ReadCorrectCount()          # start the on-device counter from zero
        for i in range(self.current_iter_, max_iter):
            this_err = self.TrainOneBatch()      # increments current_iter_ (the reference increments just before)
            if this_err is not None:             # unfused: per-step host value; fused: read every print_after below
                train_error = this_err if train_error is None else [a + b for a, b in zip(train_error, this_err)]
            if print_after > 0 and (i + 1) % print_after == 0:
                if self.fused and train_error is None:    # fused softmax output: the count lives on the device
                    train_error = [self.ReadCorrectCount()]
                if self.exchange_ is not None and hasattr(self.exchange_, "SumScalars"):
                    train_error = self.exchange_.SumScalars(train_error)
                acc = [e / (print_after * self.batch_size_ * self.num_processes_) for e in train_error]
                now = time.time()
                if self.is_root_:
                    log(f"Step {self.current_iter_} Time {now - start_t:.5g} s Train Acc : " + " ".join(f"{a:.5g}" for a in acc))
                history["train"].append((self.current_iter_, *acc))
                start_t, train_error = now, None
            if polyak_after > 0 and (i + 1) % polyak_after == 0 and (
                    (validate_after > 0 and (i + 1) % validate_after >= start_polyak_queue_val) or
                    (save_after > 0 and (i + 1) % save_after >= start_polyak_queue_save)):
                self.InsertPolyak()
            if val_dataset is not None and validate_after > 0 and (i + 1) % validate_after == 0:
                if polyak_after > 0:
                    self.LoadPolyakWeights()
                self.train_dataset_.Sync()
                this_val = self.Validate(val_dataset)
                val_error.append(this_val[lr_reduce_layer_id])
                if self.is_root_:
                    log(f"Step {self.current_iter_} Val Acc : " + " ".join(f"{v:.5g}" for v in this_val))
                history["val"].append((self.current_iter_, *this_val))
                if m.reduce_lr_factor < 1.0:
                    # `reduce && counter < max && dont_reduce_lr-- < 0` (:977-978): the post-decrement only runs when the
                    # first two hold, so the first qualifying validation arms the counter and the next one cuts the rate
                    fire = False
                    if self.CheckReduceLearningRate(val_error) and self.lr_reduce_counter_ < m.reduce_lr_max:
                        fire = dont_reduce_lr < 0
                        dont_reduce_lr -= 1
                    if fire:
                        dont_reduce_lr = m.reduce_lr_num_steps
                        self.lr_reduce_counter_ += 1
                        history["lr_reductions"] += 1
                        log(f"Learning rate reduced {self.lr_reduce_counter_} time(s).")
                        self.ReduceLearningRate(m.reduce_lr_factor)
            if checkpoint and save_after > 0 and (i + 1) % save_after == 0:
                self.train_dataset_.Sync()
                if self.is_root_:
                    self.Save()
        if checkpoint and (save_after <= 0 or max_iter % save_after != 0):
            self.train_dataset_.Sync()
            if self.is_root_:
                self.Save()
        return history

    # ---- feature extraction: src/convnet.cc:606-657 + src/datawriter.cc (one (cases, dims) float dataset per layer) -----
    def ExtractFeatures(self, dataset, layer_names, output_file):
        """Fprop(false) over the whole dataset (the last, partial batch contributes its first ``left_overs`` cases) and
        write each named layer's state, one case per row, to ``output_file``.  The net must have been allocated for this
        dataset's batch size (``AllocateMemory(True)`` is enough)."""
        from . import hdf5io
        layers = [self.GetLayerByName(n) for n in layer_names]
        dataset_size, batch_size = dataset.GetDataSetSize(), dataset.GetBatchSize()
        num_batches, left_overs = divmod(dataset_size, batch_size)
        if left_overs > 0:
            num_batches += 1
        dataset.Seek(0)
        rows = {l.GetName(): [] for l in layers}
        for k in range(num_batches):
            numcases = left_overs if (left_overs > 0 and k == num_batches - 1) else batch_size
            for l in self.layers_:
                l.ResetAddOrOverwrite()
            dataset.GetBatch(self.data_layers_)
            self.Fprop(False)
            for l in layers:
                rows[l.GetName()].append(l.GetState().ToNumpy().T[:numcases].copy())    # (N, dims): one case per row
        dataset.Sync()
        with hdf5io.File(output_file, "w") as f:
            for name, parts in rows.items():
                a = np.concatenate(parts, axis=0)
                f.WriteHDF5CPU(a, a.shape[0], a.shape[1], name)
