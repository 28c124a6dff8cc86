/* TEST INFRASTRUCTURE (oracle/seam) - abort-stubs for the cudamat entry points the reference's src/matrix.cc references but the
 * hot path never reaches and libconvnet_hip.so does not export (3-D conv, local / up-down-sample edges,
 * batch-norm, logistic / hinge losses, Adagrad / RMSProp, bounding boxes, P2P, Fermi checks).  Kept in a C file that does not see
 * the reference headers, so the stubs need no signatures. */
#include <stdio.h>
#include <stdlib.h>
#define SEAM_STUB(name)                                                                          \
  int name() {                                                                                   \
    fprintf(stderr, "libconvnet_hip.so does not provide " #name " (out of hot-path scope)\n");   \
    abort();                                                                                     \
  }
SEAM_STUB(DownSampleGemm)
SEAM_STUB(UpSampleGemm)
SEAM_STUB(ResponseNormCrossMap3DGemm)
SEAM_STUB(ResponseNormCrossMap3DUndoGemm)
SEAM_STUB(adagrad)
SEAM_STUB(rms_prop)
SEAM_STUB(apply_logistic_deriv)
SEAM_STUB(apply_logistic_grad)
SEAM_STUB(apply_relu_squash)
SEAM_STUB(apply_sigmoid)
SEAM_STUB(bn_bprop)
SEAM_STUB(bn_bprop_inplace)
SEAM_STUB(bn_grad)
SEAM_STUB(compute_cross_entropy)
SEAM_STUB(convDown3DGemm)
SEAM_STUB(convOutp3DGemm)
SEAM_STUB(convUp3DGemm)
SEAM_STUB(copy_on_device_p2p_async)
SEAM_STUB(copy_transpose_big_matrix)
SEAM_STUB(cuda_is_fermi)
SEAM_STUB(cuda_set_P2P)
SEAM_STUB(divide_elementwise)
SEAM_STUB(get_logistic_correct_normalized)
SEAM_STUB(hinge_loss_row_major)
SEAM_STUB(localDownGemm)
SEAM_STUB(localOutpGemm)
SEAM_STUB(localUpGemm)
SEAM_STUB(rectify_bounding_boxes)
