"""Edges (operators) — mirror of src/edge.{h,cc}, edge_with_weight.cc, conv_edge.cc, fc_edge.cc,
maxpool_edge.cc, avgpool_edge.cc, response_norm_edge.cc: same class names, same
ComputeUp / ComputeDown / ComputeOuter / UpdateWeights contract, same parameter slicing.

Local / one-to-one / up-down-sample / RGB->YUV edges are out of hot-path scope.
``fused`` selects the library's fused entry points (conv+bias+ReLU epilogue, one-pass bias
gradient); the unfused path issues exactly the reference's Matrix-call sequence.
"""
import math
import os

from ._lib import ConvDesc
from .matrix import Matrix
from .optimizer import Optimizer


def _divup(x, y):
    return (x + y - 1) // y


class Edge:
    can_fuse_mask = False   # ComputeDown(fuse_mask=...) supported

    @staticmethod
    def ChooseEdgeClass(edge_config):
        # src/edge.cc:19-66
        table = {"FC": FCEdge, "CONVOLUTIONAL": ConvEdge, "MAXPOOL": MaxPoolEdge, "AVERAGE_POOL": AvgPoolEdge,
                 "RESPONSE_NORM": ResponseNormEdge, "CONV_ONETOONE": ConvOneToOneEdge}
        if edge_config.edge_type not in table:
            raise SystemExit(f"Error: Undefined edge type {edge_config.edge_type} (out of hot-path scope).")
        return table[edge_config.edge_type](edge_config)

    @staticmethod
    def HasParameters(edge_config):
        return edge_config.edge_type in ("FC", "CONVOLUTIONAL", "LOCAL", "CONV_ONETOONE")

    @staticmethod
    def GetConvDesc(c):
        # src/edge.cc:83-106 — paddings are stored negated
        d = ConvDesc()
        d.num_input_channels = 0
        d.num_output_channels = 0
        d.kernel_size_y = c.kernel_size_y if c.has_kernel_size_y() else c.kernel_size
        d.kernel_size_x = c.kernel_size_x if c.has_kernel_size_x() else c.kernel_size
        d.kernel_size_t = c.kernel_size_t if c.has_kernel_size_t() else 1
        d.stride_y = c.stride_y if c.has_stride_y() else c.stride
        d.stride_x = c.stride_x if c.has_stride_x() else c.stride
        d.stride_t = c.stride_t
        d.padding_y = -(c.padding_y if c.has_padding_y() else c.padding)
        d.padding_x = -(c.padding_x if c.has_padding_x() else c.padding)
        d.padding_t = -c.padding_t
        d.num_groups = 1
        return d

    @staticmethod
    def GetNumModules(d, image_size_y, image_size_x, image_size_t):
        # src/edge.cc:108-114
        return ((image_size_y - 2 * d.padding_y - d.kernel_size_y) // d.stride_y + 1,
                (image_size_x - 2 * d.padding_x - d.kernel_size_x) // d.stride_x + 1,
                (image_size_t - 2 * d.padding_t - d.kernel_size_t) // d.stride_t + 1)

    def __init__(self, c):
        self.source_ = None
        self.dest_ = None
        self.source_node_ = c.source
        self.dest_node_ = c.dest
        self.tied_edge_name_ = c.tied_to
        self.tied_edge_ = None
        self.num_input_channels_ = 0
        self.num_output_channels_ = 0
        self.image_size_y_ = self.image_size_x_ = self.image_size_t_ = 1
        self.num_modules_y_ = self.num_modules_x_ = self.num_modules_t_ = 1
        self.mark_ = False
        self.block_backprop_ = c.block_backprop
        self.is_tied_ = bool(self.tied_edge_name_)
        self.grad_check_ = c.grad_check
        self.grad_check_num_params_ = c.grad_check_num_params
        self.grad_check_epsilon_ = list(c.grad_check_epsilon)
        self.name_ = f"{self.source_node_}:{self.dest_node_}"
        self.fused = False
        if c.source_slice or c.dest_slice:
            raise SystemExit("layer slices are out of hot-path scope")

    def GetDescription(self):
        return "Default edge."

    def SetTiedTo(self, e):
        self.tied_edge_ = e

    def SetInputChannels(self, a):
        self.num_input_channels_ = a

    def SetOutputChannels(self, a):
        self.num_output_channels_ = a
        Matrix.RegisterTempMemory(a, "Used for computing average length of incoming weight vectors.")

    def SetImageSize(self, y, x, t):
        self.image_size_y_, self.image_size_x_, self.image_size_t_ = y, x, t

    def Initialize(self):
        pass

    def SetMemory(self, p):
        pass

    def SetGradMemory(self, p):
        pass

    def GetParameterMemoryRequirement(self):
        return 0

    def ComputeOuter(self, input, deriv_output):
        pass

    def UpdateWeights(self):
        pass

    def SaveParameters(self, file):       # src/edge.cc: edges without parameters store nothing
        pass

    def LoadParameters(self, file, edge_name=None):
        pass

    def NotifyStart(self):
        pass

    def GetRMSWeight(self):
        return 0.0

    def SetSource(self, l):
        self.source_ = l

    def SetDest(self, l):
        self.dest_ = l

    def GetSource(self):
        return self.source_

    def GetDest(self):
        return self.dest_

    def GetSourceName(self):
        return self.source_node_

    def GetDestName(self):
        return self.dest_node_

    def GetSourceSliceName(self):
        return ""

    def GetDestSliceName(self):
        return ""

    def GetName(self):
        return self.name_

    def SetMark(self):
        self.mark_ = True

    def HasMark(self):
        return self.mark_

    def HasNoParameters(self):
        return True

    def GetNumModulesY(self):
        return self.num_modules_y_

    def GetNumModulesX(self):
        return self.num_modules_x_

    def GetNumModulesT(self):
        return self.num_modules_t_

    def GetTiedEdgeName(self):
        return self.tied_edge_name_

    def IsTied(self):
        return self.is_tied_

    def IsBackPropBlocked(self):
        return self.block_backprop_

    def GradCheck(self):
        return self.grad_check_

    def GradCheckNumParams(self):
        return self.grad_check_num_params_

    def GradCheckEpsilon(self):
        return list(self.grad_check_epsilon_)


class EdgeWithWeight(Edge):
    """src/edge_with_weight.{h,cc}"""

    def __init__(self, c):
        super().__init__(c)
        self.weight_optimizer_ = Optimizer.ChooseOptimizer(c.weight_optimizer)
        self.bias_optimizer_ = None if c.has_no_bias else Optimizer.ChooseOptimizer(c.bias_optimizer)
        self.initialization_ = c.initialization
        self.init_wt_ = c.init_wt
        self.init_bias_ = c.init_bias
        self.has_no_bias_ = c.has_no_bias
        self.num_grads_received_ = 0
        self.num_shares_ = 1
        self.scale_gradients_ = c.scale_gradients
        self.weights_, self.grad_weights_, self.bias_, self.grad_bias_ = Matrix(), Matrix(), Matrix(), Matrix()
        self.history_slice_ = None   # optional flat optimizer-state slice handed in by ConvNet

    def HasNoParameters(self):
        return False

    def GetWeight(self):
        return self.weights_

    def GetGradWeight(self):
        return self.grad_weights_

    def GetBias(self):
        return self.bias_

    def GetGradBias(self):
        return self.grad_bias_

    def SetTiedTo(self, e):
        if not isinstance(e, EdgeWithWeight):
            raise SystemExit(f"Error: Edge {self.GetName()} cannot be tied to edge {e.GetName()} which is not of the same type.")
        self.tied_edge_ = e
        e.num_shares_ += 1

    def GetNumGradsReceived(self):
        return self.tied_edge_.GetNumGradsReceived() if self.is_tied_ else self.num_grads_received_

    def IncrementNumGradsReceived(self):
        if self.is_tied_:
            self.tied_edge_.IncrementNumGradsReceived()
        else:
            self.num_grads_received_ += 1

    def ReduceLearningRate(self, factor):
        self.weight_optimizer_.ReduceLearningRate(factor)
        if self.bias_optimizer_:
            self.bias_optimizer_.ReduceLearningRate(factor)

    def SaveParameters(self, file):
        # src/edge_with_weight.cc:27-40: "<source>:<dest>:weight" / ":bias" + the optimizers' state under the same prefix
        if self.is_tied_:
            return
        name = f"{self.source_node_}:{self.dest_node_}:weight"
        self.weights_.WriteHDF5(file, name)
        self.weight_optimizer_.SaveParameters(file, name)
        if not self.has_no_bias_:
            name = f"{self.source_node_}:{self.dest_node_}:bias"
            self.bias_.WriteHDF5(file, name)
            self.bias_optimizer_.SaveParameters(file, name)

    def LoadParameters(self, file, edge_name=None):
        # src/edge_with_weight.cc:42-64 (optimizer state only if the optimizer has been allocated, i.e. when training)
        if self.is_tied_:
            return
        edge_name = edge_name or f"{self.source_node_}:{self.dest_node_}"
        self.weights_.ReadHDF5(file, f"{edge_name}:weight")
        if self.weight_optimizer_.IsAllocated():
            self.weight_optimizer_.LoadParameters(file, f"{edge_name}:weight")
        if not self.has_no_bias_:
            self.bias_.ReadHDF5(file, f"{edge_name}:bias")
            if self.bias_optimizer_.IsAllocated():
                self.bias_optimizer_.LoadParameters(file, f"{edge_name}:bias")

    def UpdateWeights(self, batch=None):
        # src/edge_with_weight.cc:96-106.  `batch`: a list that collects the plain fused SGD steps of a whole net for ONE launch
        # (ConvNet.UpdateWeights -> Matrix.SGDMomentumStepMulti); steps that are not plain run here as before.
        if self.is_tied_:
            return
        if self.num_grads_received_ < self.num_shares_:
            raise SystemExit("Error: Update called when all gradients were not received.")
        self.num_grads_received_ = 0
        pairs = [(self.weight_optimizer_, self.grad_weights_, self.weights_)]
        if not self.has_no_bias_:
            pairs.append((self.bias_optimizer_, self.grad_bias_, self.bias_))
        for opt, grad, param in pairs:
            item = opt.PlanFusedStep(grad, param) if batch is not None and hasattr(opt, "PlanFusedStep") else None
            if item is not None:
                batch.append(item)
            else:
                opt.Optimize(grad, param)

    def NotifyStart(self):
        self.weight_optimizer_.NotifyStart(self.weights_)
        if not self.has_no_bias_:
            self.bias_optimizer_.NotifyStart(self.bias_)

    def Initialize(self):
        # src/edge_with_weight.cc:108-143
        if self.is_tied_:
            return
        init = self.initialization_
        if init in ("DENSE_GAUSSIAN_SQRT_FAN_IN", "DENSE_GAUSSIAN"):
            self.weights_.FillWithRandn()
            init_wt = self.init_wt_
            if init == "DENSE_GAUSSIAN_SQRT_FAN_IN":
                init_wt /= math.sqrt(self.weights_.GetCols())
            self.weights_.Mult(init_wt)
        elif init in ("DENSE_UNIFORM_SQRT_FAN_IN", "DENSE_UNIFORM"):
            self.weights_.FillWithRand()
            self.weights_.Add(-0.5)
            init_wt = 2 * self.init_wt_
            if init == "DENSE_UNIFORM_SQRT_FAN_IN":
                init_wt /= math.sqrt(self.weights_.GetCols() / 3.0)
            self.weights_.Mult(init_wt)
        elif init == "CONSTANT":
            self.weights_.Set(self.init_wt_)
        else:
            raise SystemExit(f"Unknown / out-of-scope weight initialization type {init}.")
        if not self.has_no_bias_:
            self.bias_.Set(self.init_bias_)

    def GetRMSWeight(self):
        temp = Matrix()
        num_hid = self.weights_.GetRows()
        Matrix.GetTemp(num_hid, 1, temp)
        self.weights_.SqSumAxis(temp, 1, 1, 0)
        temp.Sqrt()
        return temp.Sum() / num_hid

    def _alloc_optimizers(self, rows, cols, bias_cols, hist):
        """Optimizer state: either separate matrices (reference) or slices of a flat history
        buffer laid out exactly like the parameter slice (``hist``)."""
        if hist is not None:
            hist.Reshape(rows, -1)
            hw = Matrix()
            hist.GetSlice(hw, 0, cols)
            self.weight_optimizer_.AllocateMemory(rows, cols, hw)
            if not self.has_no_bias_:
                hb = Matrix()
                hist.GetSlice(hb, cols, cols + bias_cols)
                self.bias_optimizer_.AllocateMemory(1, rows * bias_cols, hb)
        else:
            self.weight_optimizer_.AllocateMemory(rows, cols)
            if not self.has_no_bias_:
                self.bias_optimizer_.AllocateMemory(1, rows * bias_cols)


class ConvEdge(EdgeWithWeight):
    """src/conv_edge.{h,cc} (2-D; image_size_t == 1 in all target configs)."""
    can_fuse_mask = True

    def __init__(self, c):
        super().__init__(c)
        self.conv_desc_ = Edge.GetConvDesc(c)
        self.shared_bias_ = c.shared_bias

    def GetConvDesc(self):
        return self.conv_desc_

    def SetImageSize(self, y, x, t):
        super().SetImageSize(y, x, t)
        d = self.conv_desc_
        d.num_input_channels = self.num_input_channels_
        d.num_output_channels = self.num_output_channels_
        d.input_channel_end = self.num_input_channels_
        d.output_channel_end = self.num_output_channels_
        self.num_modules_y_, self.num_modules_x_, self.num_modules_t_ = Edge.GetNumModules(d, y, x, t)
        if t != 1:
            raise SystemExit("3-D convolution is out of hot-path scope")

    def GetDescription(self):
        d = self.conv_desc_
        return (f"{self.name_} Convolutional Kernel: {d.kernel_size_y}-{d.kernel_size_x}-{d.num_input_channels} : "
                f"{d.num_output_channels} Layer: {self.image_size_y_}-{self.image_size_x_} : {self.num_modules_y_}-{self.num_modules_x_}")

    def _input_size(self):
        d = self.conv_desc_
        return d.kernel_size_y * d.kernel_size_x * d.kernel_size_t * d.num_input_channels

    def _bias_locs(self):
        return 1 if self.shared_bias_ else self.num_modules_y_ * self.num_modules_x_ * self.num_modules_t_

    def GetParameterMemoryRequirement(self):
        # src/conv_edge.cc:72-78
        if self.is_tied_:
            return 0
        return self.conv_desc_.num_output_channels * (self._input_size() + (0 if self.has_no_bias_ else self._bias_locs()))

    def SetMemory(self, p):
        # src/conv_edge.cc:80-108
        if self.is_tied_:
            return
        d = self.conv_desc_
        input_size, bias_locs = self._input_size(), self._bias_locs()
        p.Reshape(d.num_output_channels, -1)
        p.GetSlice(self.weights_, 0, input_size)
        self.weights_.SetShape4D(d.num_output_channels, d.kernel_size_x, d.kernel_size_y, d.num_input_channels * d.kernel_size_t)
        if not self.has_no_bias_:
            p.GetSlice(self.bias_, input_size, input_size + bias_locs)
            self.bias_.Reshape(1, -1)

    def SetGradMemory(self, p, hist=None):
        # src/conv_edge.cc:110-136
        d = self.conv_desc_
        input_size, bias_locs = self._input_size(), self._bias_locs()
        num_locs = self.num_modules_y_ * self.num_modules_x_ * self.num_modules_t_
        if not self.is_tied_:
            p.Reshape(d.num_output_channels, -1)
            p.GetSlice(self.grad_weights_, 0, input_size)
            self.grad_weights_.SetShape4D_like(self.weights_)
            if not self.has_no_bias_:
                p.GetSlice(self.grad_bias_, input_size, input_size + bias_locs)
                self.grad_bias_.Reshape(1, -1)
                if self.shared_bias_:
                    Matrix.RegisterTempMemory(d.num_output_channels * num_locs, "shared bias")
            self._alloc_optimizers(d.num_output_channels, input_size, bias_locs, hist)

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        """src/conv_edge.cc:138-170.  ``fuse_relu`` (None = unfused reference sequence; True/False =
        fused conv+bias[+ReLU] epilogue; the caller then skips the layer's ApplyActivation)."""
        w = self.tied_edge_.GetWeight() if self.is_tied_ else self.weights_
        scale_targets = 0 if overwrite else 1
        d = self.conv_desc_
        if fuse_relu is not None and (self.has_no_bias_ or self.shared_bias_):
            b = None if self.has_no_bias_ else (self.tied_edge_.GetBias() if self.is_tied_ else self.bias_)
            Matrix.ConvUpBiasAct(input, w, b, output, d, scale_targets, fuse_relu)
            return
        Matrix.ConvUp(input, w, output, d, scale_targets)
        if not self.has_no_bias_:
            b = self.tied_edge_.GetBias() if self.is_tied_ else self.bias_
            if self.shared_bias_:
                output.Reshape(-1, d.num_output_channels)
                output.AddRowVec(b)
                output.Reshape(-1, d.num_output_channels * self.num_modules_y_ * self.num_modules_x_ * self.num_modules_t_)
            else:
                output.AddRowVec(b)

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite, fuse_mask=None):
        """src/conv_edge.cc:172-181.  ``fuse_mask=post_scale`` additionally applies the source layer's
        ReLU' (mask = its state, ``input``) and dropout' scale in the kernel epilogue."""
        w = self.tied_edge_.GetWeight() if self.is_tied_ else self.weights_
        if fuse_mask is not None:
            Matrix.ConvDownMask(deriv_output, w, input, deriv_input, self.conv_desc_, 0 if overwrite else 1, fuse_mask)
            return
        Matrix.ConvDown(deriv_output, w, deriv_input, self.conv_desc_, 0 if overwrite else 1)

    def ComputeOuter(self, input, deriv_output):
        # src/conv_edge.cc:183-245 (GEMM build: partial sums forced to one chunk, :11-17)
        dw = self.tied_edge_.GetGradWeight() if self.is_tied_ else self.grad_weights_
        batch_size = input.GetRows()
        scale_targets = 1 if self.GetNumGradsReceived() > 0 else 0
        d = self.conv_desc_
        if self.fused and self.shared_bias_ and not self.has_no_bias_:
            db = self.tied_edge_.GetGradBias() if self.is_tied_ else self.grad_bias_
            Matrix.ConvOutpBias(input, deriv_output, dw, db, d, scale_targets, self.scale_gradients_ / batch_size)
            self.IncrementNumGradsReceived()
            return
        Matrix.ConvOutp(input, deriv_output, dw, d, self.num_modules_y_, self.num_modules_x_, scale_targets,
                        self.scale_gradients_ / batch_size)
        if not self.has_no_bias_:
            db = self.tied_edge_.GetGradBias() if self.is_tied_ else self.grad_bias_
            if self.shared_bias_:
                if self.fused:
                    # one pass: (N*My*Mx, F) column sums == the reference's two-step SumRows
                    cols = deriv_output.GetCols()
                    deriv_output.Reshape(-1, d.num_output_channels)
                    deriv_output.SumRows(db, scale_targets, self.scale_gradients_ / batch_size)
                    deriv_output.Reshape(-1, cols)
                else:
                    db_temp = Matrix()
                    Matrix.GetTemp(1, deriv_output.GetCols(), db_temp)
                    deriv_output.SumRows(db_temp, 0, 1)
                    db_temp.Reshape(-1, d.num_output_channels)
                    db_temp.SumRows(db, scale_targets, self.scale_gradients_ / batch_size)
            else:
                deriv_output.SumRows(db, scale_targets, self.scale_gradients_ / batch_size)
        self.IncrementNumGradsReceived()


class FCEdge(EdgeWithWeight):
    """src/fc_edge.{h,cc}"""
    can_fuse_mask = True

    def _input_size(self):
        return self.image_size_y_ * self.image_size_x_ * self.image_size_t_ * self.num_input_channels_

    def GetParameterMemoryRequirement(self):
        if self.is_tied_:
            return 0
        return self.num_output_channels_ * (self._input_size() + (0 if self.has_no_bias_ else 1))

    def GetDescription(self):
        return f"{self.name_} Fully Connected :{self.image_size_y_}-{self.image_size_x_}-{self.num_input_channels_}:{self.num_output_channels_}"

    def SetMemory(self, p):
        if self.is_tied_:
            return
        input_size = self._input_size()
        p.Reshape(self.num_output_channels_, -1)
        p.GetSlice(self.weights_, 0, input_size)
        if not self.has_no_bias_:
            p.GetSlice(self.bias_, input_size, input_size + 1)
            self.bias_.Reshape(1, -1)

    def SetGradMemory(self, p, hist=None):
        if self.is_tied_:
            return
        input_size = self._input_size()
        p.Reshape(self.num_output_channels_, -1)
        p.GetSlice(self.grad_weights_, 0, input_size)
        if not self.has_no_bias_:
            p.GetSlice(self.grad_bias_, input_size, input_size + 1)
            self.grad_bias_.Reshape(1, -1)
        self._alloc_optimizers(self.num_output_channels_, input_size, 1, hist)

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        # src/fc_edge.cc:51-61
        w = self.tied_edge_.GetWeight() if self.is_tied_ else self.weights_
        scale_targets = 0 if overwrite else 1
        if fuse_relu is not None:
            b = None if self.has_no_bias_ else (self.tied_edge_.GetBias() if self.is_tied_ else self.bias_)
            Matrix.DotBiasAct(input, w, b, output, scale_targets, 1, False, True, fuse_relu)
            return
        Matrix.Dot(input, w, output, scale_targets, 1, False, True)
        if not self.has_no_bias_:
            output.AddRowVec(self.tied_edge_.GetBias() if self.is_tied_ else self.bias_)

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite, fuse_mask=None):
        # src/fc_edge.cc:63-68
        w = self.tied_edge_.GetWeight() if self.is_tied_ else self.weights_
        if fuse_mask is not None:
            Matrix.DotMask(deriv_output, w, input, deriv_input, 0 if overwrite else 1, 1, fuse_mask)
            return
        Matrix.Dot(deriv_output, w, deriv_input, 0 if overwrite else 1, 1)

    def ComputeOuter(self, input, deriv_output):
        # src/fc_edge.cc:70-81
        dw = self.tied_edge_.GetGradWeight() if self.is_tied_ else self.grad_weights_
        scale_targets = 1 if self.GetNumGradsReceived() > 0 else 0
        batch_size = input.GetRows()
        Matrix.Dot(deriv_output, input, dw, scale_targets, self.scale_gradients_ / batch_size, True, False)
        if not self.has_no_bias_:
            db = self.tied_edge_.GetGradBias() if self.is_tied_ else self.grad_bias_
            deriv_output.SumRows(db, scale_targets, self.scale_gradients_ / batch_size)
        self.IncrementNumGradsReceived()


class ConvOneToOneEdge(FCEdge):
    """src/conv_onetoone_edge.{h,cc}: a 1x1 convolution (network-in-network layer) = the FC GEMMs on the
    (N*X*Y, C) view of the same CHWN bytes, so pixel and image together form the contiguous "image" axis of the
    gather-GEMM kernels.  Everything but the reshapes is FCEdge's."""

    def SetImageSize(self, y, x, t):
        # src/conv_onetoone_edge.cc:8-13
        Edge.SetImageSize(self, y, x, t)
        self.num_modules_y_, self.num_modules_x_, self.num_modules_t_ = y, x, t

    def _input_size(self):
        return self.num_input_channels_

    def GetDescription(self):
        return (f"{self.name_}  One-to-One Convolutional Kernel: {self.num_input_channels_} : {self.num_output_channels_} Layer: "
                f"{self.image_size_y_}-{self.image_size_x_}-{self.num_input_channels_} : {self.num_modules_y_}-{self.num_modules_x_}-"
                f"{self.num_output_channels_}")

    class _Flat:
        """with-block: view activations as (N*X*Y, channels) and restore (batch, -1) on exit (:58-59,72-73)."""

        def __init__(self, *pairs):
            self.pairs = [(m, ch) for m, ch in pairs if m is not None]

        def __enter__(self):
            self.rows = [m.GetRows() for m, _ in self.pairs]
            for m, ch in self.pairs:
                m.Reshape(-1, ch)

        def __exit__(self, *exc):
            for (m, _), rows in zip(self.pairs, self.rows):
                m.Reshape(rows, -1)

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        with self._Flat((input, self.num_input_channels_), (output, self.num_output_channels_)):
            FCEdge.ComputeUp(self, input, output, overwrite, train, fuse_relu)

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite, fuse_mask=None):
        with self._Flat((deriv_output, self.num_output_channels_), (deriv_input, self.num_input_channels_),
                        (input if fuse_mask is not None else None, self.num_input_channels_)):
            FCEdge.ComputeDown(self, deriv_output, input, output, deriv_input, overwrite, fuse_mask)

    def ComputeOuter(self, input, deriv_output):
        # scale_gradients / batch_size uses the layer's batch (rows before the reshape), :92-104
        batch_size = input.GetRows()
        dw = self.tied_edge_.GetGradWeight() if self.is_tied_ else self.grad_weights_
        scale_targets = 1 if self.GetNumGradsReceived() > 0 else 0
        with self._Flat((input, self.num_input_channels_), (deriv_output, self.num_output_channels_)):
            Matrix.Dot(deriv_output, input, dw, scale_targets, self.scale_gradients_ / batch_size, True, False)
            if not self.has_no_bias_:
                db = self.tied_edge_.GetGradBias() if self.is_tied_ else self.grad_bias_
                deriv_output.SumRows(db, scale_targets, self.scale_gradients_ / batch_size)
        self.IncrementNumGradsReceived()


class _PoolEdge(Edge):
    def __init__(self, c):
        super().__init__(c)
        self.conv_desc_ = Edge.GetConvDesc(c)

    def GetConvDesc(self):
        return self.conv_desc_

    def SetImageSize(self, y, x, t):
        # src/maxpool_edge.cc:13-25
        super().SetImageSize(y, x, t)
        d = self.conv_desc_
        d.num_input_channels = self.num_input_channels_
        d.num_output_channels = self.num_output_channels_
        d.input_channel_end = self.num_input_channels_
        d.output_channel_end = self.num_output_channels_
        if d.kernel_size_y <= 0:
            d.kernel_size_y = y
        if d.kernel_size_x <= 0:
            d.kernel_size_x = x
        if d.kernel_size_t <= 0:
            d.kernel_size_t = t
        self.num_modules_y_, self.num_modules_x_, self.num_modules_t_ = Edge.GetNumModules(d, y, x, t)


_POOL_MASK = os.environ.get("CONVNET_POOL_MASK", "1") != "0"   # A/B switch (tools/profile_round.sh): 0 = the reference's call pair on the fused path too


class MaxPoolEdge(_PoolEdge):
    """src/maxpool_edge.{h,cc}.  With the host's fused entry points on (ConvNet(fused=True) sets ``fused``) the forward pass also records
    the window masks (include/convnet_hip.h: MaxPoolMask) and the backward pass routes the derivatives from them alone — it reads neither
    the layer's input (1.19 GB for AlexNet's pool1) nor its maxima.  Bit-identical to the reference's call pair; geometries without a mask
    kernel, and a ComputeDown that is handed other matrices than the ComputeUp before it, take the reference's calls."""
    can_fuse_mask = True

    def __init__(self, c):
        super().__init__(c)
        self.fused = False
        self.mask_ = None
        self.mask_for_ = None   # (input data pointer, output data pointer, batch) of the ComputeUp that wrote mask_

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        if not overwrite:
            raise SystemExit(" In MaxPoolEdge::ComputeUp() : some other layer is writing to this maxpool layer's output as well. Not implemented.")
        self.mask_for_ = None
        if self.fused and train and _POOL_MASK:
            need = (output.GetRows(), (output.GetCols() + 1) // 2)
            if self.mask_ is None or (self.mask_.GetRows(), self.mask_.GetCols()) != need:
                self.mask_ = Matrix()
                self.mask_.AllocateGPUMemory(need[0], need[1], "maxpool mask")
            if Matrix.ConvMaxPoolMask(input, output, self.mask_, self.conv_desc_):
                self.mask_for_ = (input.mat_.data_device, output.mat_.data_device, output.GetRows())
                return
        Matrix.ConvMaxPool(input, output, self.conv_desc_)

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite, fuse_mask=None):
        relu = fuse_mask is not None and fuse_mask == 1.0
        if self.mask_for_ is not None and self.mask_for_ == (input.mat_.data_device, output.mat_.data_device, output.GetRows()) and (overwrite or not relu):
            Matrix.ConvMaxPoolUndoMask(deriv_output, self.mask_, deriv_input, self.conv_desc_, 0 if overwrite else 1, relu)
            return
        if relu:
            Matrix.ConvMaxPoolUndoRelu(input, deriv_output, output, deriv_input, self.conv_desc_, 0 if overwrite else 1)
            return
        Matrix.ConvMaxPoolUndo(input, deriv_output, output, deriv_input, self.conv_desc_, 0 if overwrite else 1)


class AvgPoolEdge(_PoolEdge):
    """src/avgpool_edge.{h,cc}"""

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        if not overwrite:
            raise SystemExit(" In AvgPoolEdge::ComputeUp() : some other layer is writing to this layer's output as well. Not implemented.")
        Matrix.ConvAvgPool(input, output, self.conv_desc_)

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite):
        Matrix.ConvAvgPoolUndo(deriv_output, deriv_input, self.conv_desc_, 0 if overwrite else 1)


class ResponseNormEdge(Edge):
    """src/response_norm_edge.{h,cc}"""

    def __init__(self, c):
        super().__init__(c)
        self.num_filters_response_norm_ = 0
        self.blocked_ = c.response_norm_in_blocks
        self.add_scale_ = c.add_scale
        self.pow_scale_ = c.pow_scale
        self.frac_of_filters_response_norm_ = c.frac_of_filters_response_norm

    def SetImageSize(self, y, x, t):
        super().SetImageSize(y, x, t)
        self.num_modules_y_, self.num_modules_x_, self.num_modules_t_ = y, x, t
        # (int) truncation of a *float* product, as in C++ (src/response_norm_edge.cc:37-38)
        import numpy as np
        self.num_filters_response_norm_ = int(np.float32(self.frac_of_filters_response_norm_) * np.float32(self.num_input_channels_))

    def ComputeUp(self, input, output, overwrite, train=True, fuse_relu=None):
        # fuse_relu=True: the destination layer's ReLU (layer.cc:549) is applied by the same kernel
        Matrix.ConvResponseNormCrossMap(input, output, self.num_input_channels_, self.num_filters_response_norm_,
                                        self.add_scale_, self.pow_scale_, self.blocked_, relu=bool(fuse_relu))

    def ComputeDown(self, deriv_output, input, output, deriv_input, overwrite):
        Matrix.ConvResponseNormCrossMapUndo(deriv_output, input, output, deriv_input, self.num_input_channels_,
                                            self.num_filters_response_norm_, self.add_scale_, self.pow_scale_, self.blocked_)
