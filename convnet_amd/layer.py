"""Layers — mirror of src/layer.{h,cc} for the hot path: Linear / ReLU / Softmax layers with binary
dropout, CE loss and the classification metric.  Batch-norm, logistic, slices and the
model-parallel state copies are out of scope."""
from .loss_functions import LossFunction
from .matrix import Matrix


class Layer:
    @staticmethod
    def ChooseLayerClass(config):
        # src/layer.cc:8-32
        a = config.activation
        if a == "LINEAR":
            return LinearLayer(config)
        if a == "RECTIFIED_LINEAR":
            return ReLULayer(config)
        if a == "SOFTMAX":
            return SoftmaxLayer(config)
        raise SystemExit(f"Undefined layer type {a} (out of hot-path scope).")

    def __init__(self, config):
        self.name_ = config.name
        self.num_channels_ = config.num_channels
        self.is_input_ = True
        self.is_output_ = True
        self.dropprob_ = config.dropprob
        self.dropout_scale_up_at_train_time_ = True
        self.gaussian_dropout_ = config.gaussian_dropout
        self.image_size_y_ = config.image_size_y
        self.image_size_x_ = config.image_size_x
        self.image_size_t_ = config.image_size_t
        self.store_dropout_noise_ = self.dropprob_ > 0
        self.loss_ = None
        self.performance_ = None
        self.loss_function_ = config.loss_function
        self.performance_metric_ = config.performance_metric
        self.loss_function_weight_ = config.loss_function_weight
        self.has_tied_data_ = bool(config.tied_data)
        self.incoming_edge_ = []
        self.outgoing_edge_ = []
        self.state_ = Matrix()
        self.deriv_ = Matrix()
        self.data_ = Matrix()
        self.dropout_noise_ = Matrix()
        self.add_or_overwrite_state_ = True
        self.add_or_overwrite_deriv_ = True
        if config.batch_normalize or config.layer_slice or self.gaussian_dropout_:
            raise SystemExit("batch_normalize / layer_slice / gaussian_dropout are out of hot-path scope")

    # ---- graph ----------------------------------------------------------------------------------------
    def AddIncoming(self, e):
        self.is_input_ = False
        self.incoming_edge_.append(e)

    def AddOutgoing(self, e):
        self.is_output_ = False
        self.outgoing_edge_.append(e)

    def GetName(self):
        return self.name_

    def GetNumChannels(self, slice_=""):
        return self.num_channels_

    def IsInput(self):
        return self.is_input_

    def IsOutput(self):
        return self.is_output_

    def GetSizeY(self):
        return self.image_size_y_

    def GetSizeX(self):
        return self.image_size_x_

    def GetSizeT(self):
        return self.image_size_t_

    def SetSize(self, y, x, t):
        self.image_size_y_, self.image_size_x_, self.image_size_t_ = y, x, t

    def GetState(self, slice_=""):
        return self.state_

    def GetDeriv(self, slice_=""):
        return self.deriv_

    def GetData(self):
        return self.data_

    # add-or-overwrite bookkeeping: src/layer.cc:307-332
    def AddOrOverwriteState(self, slice_=""):
        v = self.add_or_overwrite_state_
        self.add_or_overwrite_state_ = False
        return v

    def AddOrOverwriteDeriv(self, slice_=""):
        v = self.add_or_overwrite_deriv_
        self.add_or_overwrite_deriv_ = False
        return v

    def ResetAddOrOverwrite(self):
        self.add_or_overwrite_state_ = True
        self.add_or_overwrite_deriv_ = True

    def NotifyStart(self):
        pass

    # ---- memory: src/layer.cc:252-288 ----------------------------------------------------------------
    def AllocateMemory(self, batch_size):
        num_pixels = self.image_size_y_ * self.image_size_x_ * self.image_size_t_
        self.state_.AllocateGPUMemory(batch_size, num_pixels * self.num_channels_, self.name_ + " state")
        self.deriv_.AllocateGPUMemory(batch_size, num_pixels * self.num_channels_, self.name_ + " deriv")
        for m in (self.state_, self.deriv_):
            m.SetShape4D(batch_size, self.image_size_x_, self.image_size_y_, self.num_channels_ * self.image_size_t_)
        if self.is_input_:
            self.store_dropout_noise_ = False
        if self.store_dropout_noise_:
            self.dropout_noise_.AllocateGPUMemory(batch_size, num_pixels * self.num_channels_, self.name_ + " dropout")
        if self.is_output_:
            self.loss_ = LossFunction.ChooseLossFunction(self.loss_function_)
            self.performance_ = LossFunction.ChooseLossFunction(self.performance_metric_)

    # ---- activation / dropout ----------------------------------------------------------------------------
    def ApplyActivation(self):
        raise NotImplementedError

    def ApplyDerivativeOfActivation(self):
        raise NotImplementedError

    def ApplyDropout(self, train):
        if train:
            self.ApplyDropoutAtTrainTime()
        else:
            self.ApplyDropoutAtTestTime()

    def ApplyDropoutAtTrainTime(self):
        # src/layer.cc:367-397
        if self.dropprob_ > 0:
            scale = 1.0 / (1 - self.dropprob_) if self.dropout_scale_up_at_train_time_ else 1.0
            if self.store_dropout_noise_:
                self.dropout_noise_.SampleBernoulli(1 - self.dropprob_)
                self.dropout_noise_.Mult(scale)
                self.state_.Mult(self.dropout_noise_)
            else:
                self.state_.Dropout(self.dropprob_, 0, scale)

    def ApplyDerivativeofDropout(self):
        # src/layer.cc:399-413
        if self.dropprob_ > 0:
            if self.store_dropout_noise_:
                self.deriv_.Mult(self.dropout_noise_)
            elif self.dropout_scale_up_at_train_time_:
                self.deriv_.Mult(1.0 / (1 - self.dropprob_))

    def ApplyDropoutAtTestTime(self):
        if self.dropprob_ > 0 and not self.dropout_scale_up_at_train_time_:
            self.state_.Mult(1 - self.dropprob_)

    # ---- loss ---------------------------------------------------------------------------------------
    def GetPerformanceMetric(self):
        return self.performance_.GetLoss(self.state_, self.data_)

    def ComputeDeriv(self):
        self.loss_.GetLossDerivative(self.state_, self.data_, self.deriv_)
        if self.loss_function_weight_ != 1.0:
            self.deriv_.Mult(self.loss_function_weight_)

    def GetLoss(self):
        return self.loss_function_weight_ * self.loss_.GetLoss(self.state_, self.data_)


class LinearLayer(Layer):
    is_relu = False

    def ApplyActivation(self):
        pass  # linear: nothing to do (src/layer.cc:530-532)

    def ApplyDerivativeOfActivation(self):
        pass

    def AllocateMemory(self, batch_size):
        super().AllocateMemory(batch_size)
        num_pixels = self.image_size_y_ * self.image_size_x_ * self.image_size_t_
        if self.is_output_:
            self.data_.AllocateGPUMemory(batch_size, num_pixels * self.num_channels_, self.name_ + " data")


class ReLULayer(LinearLayer):
    is_relu = True

    def __init__(self, config):
        super().__init__(config)
        self.store_dropout_noise_ = False  # src/layer.cc:544-547: ReLU' zeroes dropped units

    def ApplyActivation(self):
        self.state_.LowerBound(0)

    def ApplyDerivativeOfActivation(self):
        self.deriv_.ApplyDerivativeOfReLU(self.state_)


class SoftmaxLayer(Layer):
    is_relu = False

    def AllocateMemory(self, batch_size):
        super().AllocateMemory(batch_size)
        if self.is_output_:
            self.data_.AllocateGPUMemory(batch_size, 1, self.name_ + " data")
        Matrix.RegisterTempMemory(batch_size)

    def ApplyActivation(self):
        self.state_.ApplySoftmax()

    def ApplyDerivativeOfActivation(self):
        raise SystemExit("Back prop through softmax is not implemented.")
