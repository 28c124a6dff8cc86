"""Loss functions — mirror of src/loss_functions.{h,cc} for the output layers the target configs use."""
from .matrix import Matrix


class LossFunction:
    @staticmethod
    def ChooseLossFunction(lf):
        # src/loss_functions.cc:5-36
        table = {"SQUARED_ERROR": SquaredError, "CROSS_ENTROPY_MULTINOMIAL": CrossEntropyMultinomial,
                 "CLASSIFICATION_MULTINOMIAL": ClassificationMultinomial}
        if lf not in table:
            raise SystemExit(f"Unknown loss function {lf}")
        return table[lf]()


class SquaredError(LossFunction):
    def GetLoss(self, y, t):
        temp = Matrix()
        Matrix.GetTemp(t.GetRows(), t.GetCols(), temp)
        y.Subtract(t, temp)
        norm = temp.EuclidNorm()
        return 0.5 * norm * norm

    def GetLossDerivative(self, y, t, dLbydy):
        y.Subtract(t, dLbydy)


class CrossEntropyMultinomial(LossFunction):
    def GetLoss(self, y, t):
        temp = Matrix()
        Matrix.GetTemp(t.GetRows(), 1, temp)
        Matrix.SoftmaxCE(y, t, temp)
        return temp.Sum()

    def GetLossDerivative(self, y, t, dLbydy):
        Matrix.SoftmaxCEDeriv(y, t, dLbydy)


class ClassificationMultinomial(LossFunction):
    def GetLoss(self, y, t):
        temp = Matrix()
        Matrix.GetTemp(t.GetRows(), 1, temp)
        Matrix.SoftmaxCorrect(y, t, temp)
        return temp.Sum()

    def GetLossDerivative(self, y, t, dLbydy):
        dLbydy.Set(0)
