"""Optimizers — mirror of src/optimizer.{h,cc} for the hot path (SGD + momentum, the only optimizer
the target configs use; Adagrad/RMSProp/LBFGS are out of scope)."""
import ctypes
import math

import numpy as np

from .matrix import Matrix


_libm = None


def _expf(x):
    """The C library's expf — the function the reference's `exp(float)` resolves to (numpy's float32 exp is a different
    implementation and may differ in the last bit).  libm is located at first use; on a platform without one the float32 numpy
    value is used."""
    global _libm
    if _libm is None:
        import ctypes.util
        try:
            lm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
            lm.expf.restype, lm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
            _libm = lm
        except (OSError, AttributeError):
            _libm = False
    if _libm:
        return np.float32(_libm.expf(float(x)))
    return np.exp(np.float32(x), dtype=np.float32)


class Optimizer:
    @staticmethod
    def ChooseOptimizer(config):
        # src/optimizer.cc:8-29
        if config.optimizer_type == "STOCHASTIC_GRADIENT_DESCENT":
            return SGDOptimizer(config)
        raise SystemExit(f"Undefined optimizer {config.optimizer_type} (only SGD is on the hot path).")

    def __init__(self, c):
        self.epsilon_decay_type_ = c.epsilon_decay
        self.epsilon_ = c.epsilon
        self.minimum_epsilon_ = c.minimum_epsilon
        self.decay_factor_ = c.decay_factor
        self.epsilon_decay_timescale_ = c.epsilon_decay_timescale
        self.start_optimization_after_ = c.start_optimization_after
        self.l2_decay_ = c.l2_decay
        self.weight_norm_limit_ = c.weight_norm_limit
        self.weight_norm_constraint_ = c.weight_norm_constraint
        self.step_ = 0
        if c.shared_prior:
            raise SystemExit("shared_prior is out of scope")

    def ReduceLearningRate(self, factor):
        self.epsilon_ *= factor

    def ApplyConstraints(self, parameter):
        # src/optimizer.cc:75-81 (axis=1: per output unit)
        if self.weight_norm_constraint_ > 0:
            parameter.NormLimitByAxis(1, self.weight_norm_constraint_, True)
        elif self.weight_norm_limit_ > 0:
            parameter.NormLimitByAxis(1, self.weight_norm_limit_, False)

    def GetDecayedEpsilon(self):
        # src/optimizer.cc:83-104, evaluated in the reference's types: `float f`, expf, float arithmetic; only EXPONENTIAL_STEP
        # goes through double (C++11 pow(float, int) promotes both) and is rounded to float once.  Intentional deviation: for decay
        # type NONE with a timescale > 0 the reference's guard (it compares the TIMESCALE with the enum NONE, :85-86) enters the
        # branch and exits with "Unknown epsilon decay rule"; here that combination just means "no decay".
        f32 = np.float32
        eps = f32(self.epsilon_)
        ts = self.epsilon_decay_timescale_
        if ts > 0 and self.epsilon_decay_type_ != "NONE":
            f = f32(f32(self.step_) / f32(ts))
            t = self.epsilon_decay_type_
            if t == "EXPONENTIAL":
                eps = f32(f32(self.epsilon_) * _expf(-f))
            elif t == "INVERSE_T":
                eps = f32(f32(self.epsilon_) / f32(f32(1) + f))
            elif t == "LINEAR":
                eps = f32(f32(f32(self.epsilon_) * f32(f32(1) - f)) + f32(f32(self.minimum_epsilon_) * f)) if f < 1 else f32(self.minimum_epsilon_)
            elif t == "EXPONENTIAL_STEP":
                eps = f32(float(f32(self.epsilon_)) * math.pow(float(f32(self.decay_factor_)), self.step_ // ts))
            else:
                raise SystemExit("Unknown epsilon decay rule.")
        return float(max(eps, f32(self.minimum_epsilon_)))

    def NotifyStart(self, parameter):
        pass

    def AllocateMemory(self, rows, cols):
        pass

    def IsAllocated(self):
        return False

    def LoadParameters(self, file, prefix):     # src/optimizer.cc:110-111: the base class stores nothing
        pass

    def SaveParameters(self, file, prefix):
        pass


class SGDOptimizer(Optimizer):
    def __init__(self, c):
        super().__init__(c)
        self.gradient_clip_ = c.gradient_clip
        self.initial_momentum_ = c.initial_momentum
        self.final_momentum_ = c.final_momentum
        self.momentum_transition_timescale_ = c.momentum_transition_timescale
        self.nesterov_momentum_ = c.nesterov_momentum
        self.gradient_history_ = Matrix()
        self.fused = False

    def AllocateMemory(self, rows, cols, storage=None):
        """``storage``: optional Matrix slice of a flat history buffer (the reference allocates one
        matrix per tensor, src/optimizer.cc:131-134; a flat buffer makes the state contiguous)."""
        if storage is not None:
            self.gradient_history_ = storage
            self.gradient_history_.Reshape(rows, cols)
        else:
            self.gradient_history_.AllocateGPUMemory(rows, cols, "optimizer")
        self.gradient_history_.Set(0.0)

    def IsAllocated(self):
        return self.gradient_history_.GetNumEls() > 0

    def LoadParameters(self, file, prefix):
        # src/optimizer.cc:138-146
        self.gradient_history_.ReadHDF5(file, f"{prefix}_gradient_history")
        self.step_ = file.ReadHDF5IntAttr(f"{prefix}_step", self.step_)

    def SaveParameters(self, file, prefix):
        # src/optimizer.cc:148-156
        self.gradient_history_.WriteHDF5(file, f"{prefix}_gradient_history")
        file.WriteHDF5IntAttr(f"{prefix}_step", self.step_)

    def GetMomentum(self):
        # src/optimizer.cc:158-165 in float, like the reference (expf of a float argument)
        f32 = np.float32
        if self.momentum_transition_timescale_ > 0:
            x = f32(-f32(self.step_) / f32(self.momentum_transition_timescale_))
            return float(f32(f32(self.initial_momentum_) + f32(f32(f32(self.final_momentum_) - f32(self.initial_momentum_)) * f32(f32(1) - _expf(x)))))
        return self.final_momentum_

    def NotifyStart(self, parameter):
        if self.nesterov_momentum_:
            self.gradient_history_.Mult(self.GetMomentum())
            parameter.Add(self.gradient_history_, -1)

    def PlanFusedStep(self, gradient, parameter):
        """The plain fused step of Optimize as DATA — (gradient, parameter, history, l2, clip, epsilon, momentum) for
        Matrix.SGDMomentumStepMulti — with the step counter advanced exactly as Optimize would; None when this optimizer's step is not the
        plain one (unfused host, Nesterov, a norm limit / constraint, still before start_optimization_after): the caller runs Optimize."""
        if (not self.fused or self.nesterov_momentum_ or self.weight_norm_constraint_ > 0 or self.weight_norm_limit_ > 0 or
                self.step_ < self.start_optimization_after_):
            return None
        item = (gradient, parameter, self.gradient_history_, self.l2_decay_, self.gradient_clip_, self.GetDecayedEpsilon(), self.GetMomentum())
        self.step_ += 1
        return item

    def Optimize(self, gradient, parameter):
        # src/optimizer.cc:174-200
        if self.step_ >= self.start_optimization_after_:
            epsilon = self.GetDecayedEpsilon()
            if self.fused and not self.nesterov_momentum_:
                if self.weight_norm_constraint_ > 0 or self.weight_norm_limit_ > 0:
                    con = self.weight_norm_constraint_ > 0
                    Matrix.SGDMomentumStepNormLimit(gradient, parameter, self.gradient_history_, self.l2_decay_, self.gradient_clip_,
                                                    epsilon, self.GetMomentum(),
                                                    self.weight_norm_constraint_ if con else self.weight_norm_limit_, con)
                else:
                    Matrix.SGDMomentumStep(gradient, parameter, self.gradient_history_, self.l2_decay_,
                                           self.gradient_clip_, epsilon, self.GetMomentum())
                self.step_ += 1
                return
            else:
                if self.l2_decay_ > 0:
                    gradient.Add(parameter, self.l2_decay_)
                if self.gradient_clip_ > 0:
                    gradient.UpperBoundMod(self.gradient_clip_)
                gradient.Mult(epsilon)
                if not self.nesterov_momentum_:
                    self.gradient_history_.Mult(self.GetMomentum())
                self.gradient_history_.Add(gradient)
                if self.nesterov_momentum_:
                    parameter.Add(gradient, -1)
                else:
                    parameter.Add(self.gradient_history_, -1)
            self.ApplyConstraints(parameter)
        self.step_ += 1
