"""GradChecker — mirror of src/grad_check.{h,cc} / apps/run_grad_check.cc: central-difference check of
the analytic weight and bias gradients of every edge flagged ``grad_check: true`` in the pbtxt.

Same procedure: random-fill the input layers (FillWithRandn) and targets (FillWithRand -> label 0),
Fprop(false); ComputeDeriv(); Bprop(); then for the first ``grad_check_num_params`` weights and
biases perturb by +-epsilon, re-run Fprop(false), and compare
(L(w+e) - L(w-e)) / (2 e batch) with the analytic value; an edge passes if the mean of
|a-n| / |(a+n)/2| over non-zero entries is < 0.01 for ANY epsilon (src/grad_check.cc:37-75).
The HDF5 dump of the arrays is replaced by the returned dict (HDF5 output is a §8f "next" row)."""
import numpy as np

from .convnet import ConvNet
from .edge import EdgeWithWeight

_f = np.float32   # the reference does all of this arithmetic in `float`


class GradChecker(ConvNet):
    def GetLoss(self):
        # src/grad_check.cc:10-18
        for l in self.layers_:
            l.ResetAddOrOverwrite()
        self.Fprop(False)
        return [l.GetLoss() for l in self.output_layers_]

    def ComputeNumericGrad(self, w, epsilon, max_params):
        # src/grad_check.cc:20-35
        num_params = w.GetNumEls()
        if 0 < max_params < num_params:
            num_params = max_params
        out = []
        for i in range(num_params):
            val = w.ReadValue(i)
            w.WriteValue(i, float(_f(val) + _f(epsilon)))
            e1 = self.GetLoss()
            w.WriteValue(i, float(_f(val) - _f(epsilon)))
            e2 = self.GetLoss()
            out.append(float((_f(e1[0]) - _f(e2[0])) / (_f(self.batch_size_ * 2) * _f(epsilon))))
            w.WriteValue(i, val)
        return out

    def GradCheck(self, w, eps_values, num_params, analytical_g):
        # src/grad_check.cc:37-75 (including its quirk: diff_sum carries over between epsilons)
        diff_sum, non_zero, test_pass, numerical = _f(0.0), 0, False, {}
        for eps in eps_values:
            if test_pass:
                break
            this_num = self.ComputeNumericGrad(w, eps, num_params)
            for k in range(num_params):
                diff = _f(analytical_g[k]) - _f(this_num[k])
                scale = (_f(analytical_g[k]) + _f(this_num[k])) / _f(2)
                if not (scale == 0 and diff == 0):
                    with np.errstate(divide="ignore", invalid="ignore"):
                        diff_sum = _f(diff_sum + abs(diff / scale))
                    non_zero += 1
            # grad_check.cc:59: float division, 0/0 = NaN when every entry is exactly zero — and NaN < 0.01 is false, so an
            # all-zero (dead unit) gradient is reported FAILED by the reference; kept
            diff_sum = _f(diff_sum / _f(non_zero)) if non_zero else _f("nan")
            numerical[eps] = (this_num, diff_sum)
            if diff_sum < 0.01:
                test_pass = True
        return test_pass, numerical

    def Run(self, fixed_batch=False):
        """Returns {edge name: {"weights": (passed, analytical, numerical), "bias": (...)}}.
        ``fixed_batch``: differentiate at the data handler's next batch instead of the reference's random fill, so that two
        back-ends can be checked at the same point (tests; the reference's own flow is the default)."""
        for l in self.layers_:
            l.ResetAddOrOverwrite()
        if fixed_batch:
            self.GetBatch(self.train_dataset_)
        for l in self.data_layers_:
            if fixed_batch:
                break
            if l.IsInput():
                l.GetState().FillWithRandn()
            else:
                l.GetData().FillWithRand()   # (int)label == 0, as in the reference
        self.Fprop(False)
        self.ComputeDeriv()
        self.Bprop()
        results = {}
        for ed in self.edges_:
            if not ed.GradCheck() or not isinstance(ed, EdgeWithWeight):
                continue
            eps_values = ed.GradCheckEpsilon()
            res = {}
            for what, grad, param in (("weights", ed.GetGradWeight(), ed.GetWeight()), ("bias", ed.GetGradBias(), ed.GetBias())):
                g = grad.ToNumpy().reshape(-1)
                n = min(ed.GradCheckNumParams(), g.size)
                passed, numerical = self.GradCheck(param, eps_values, n, [float(x) for x in g[:n]])
                res[what] = (passed, g[:n].copy(), numerical)
            results[ed.GetName()] = res
        return results
