"""Validation path with the per-pixel work on the device (drop-in for the whole-image branch of the reference's
tools/engine/evaluator.py Evaluator + train/eval.py SegEvaluator).

Reference, per validation image (evaluator.py:205-225, 297-318; eval.py:17-28): normalise on the host, forward, exp() of the
(19, 1024, 2048) fp32 score map, copy 159 MB to the host, np.argmax, hist_info on the host.  Here: the image is normalised on
the device, the network runs from the static-plan engine in class-map mode (the final x8 up-sample and the arg-max are one
launch writing a 2 MB uint8 map), and the confusion histogram is accumulated on the device; the host reads 19 x 19 counts at
the end of the run.  Multi-scale / sliding-window / flip evaluation (evaluator.py:227-295) need cv2 resampling of the input
and are out of scope (config_train.py:67-68 uses a single scale without flip)."""
import numpy as np
import torch

from . import engine
from .metric import HistAccumulator, compute_score


class SegEvaluator:
    def __init__(self, network, class_num, image_mean, image_std, image_shape=(1024, 2048), dtype=torch.bfloat16, device="cuda"):
        self.class_num = class_num
        self.device = torch.device(device)
        self.image_mean = torch.tensor(np.asarray(image_mean, dtype=np.float32), device=self.device).view(1, 3, 1, 1)
        self.image_std = torch.tensor(np.asarray(image_std, dtype=np.float32), device=self.device).view(1, 3, 1, 1)
        H, W = image_shape
        self.val_func = network.eval()
        self.engine = engine.InferenceEngine(self.val_func, (1, 3, H, W), dtype=dtype, output="classes")
        self.acc = HistAccumulator(class_num, self.device)

    def process_image(self, img):
        """HWC uint8 (numpy or tensor, RGB like the reference after its BGR->RGB flip) -> normalised (1, 3, H, W) fp32 on the
        device: tools/utils/img_utils.py:178-184 normalize + the transpose of evaluator.py:346."""
        t = torch.as_tensor(img)
        t = t.to(self.device, non_blocking=True).permute(2, 0, 1).unsqueeze(0).float()
        return (t / 255.0 - self.image_mean) / self.image_std

    def val_func_process(self, input_data):
        """(1, 3, H, W) normalised image -> (H, W) uint8 class map on the device (argmax(exp(score)) = argmax(score))."""
        return self.engine(input_data)[0]

    def whole_eval(self, img, output_size=None, input_size=None):
        assert output_size is None and input_size is None, "resized / padded evaluation needs cv2 (out of scope)"
        return self.val_func_process(self.process_image(img))

    def func_per_iteration(self, data):
        """data: {'data': HWC uint8 image, 'label': (H, W) labels}; accumulates on the device, returns the class map."""
        pred = self.whole_eval(data['data'])
        label = torch.as_tensor(data['label']).to(self.device)
        self.acc.add(pred, label.contiguous())
        return pred

    def compute_metric(self):
        hist, labeled, correct = self.acc.result()
        iu, mean_IU, mean_IU_no_back, mean_pixel_acc = compute_score(hist, correct, labeled)
        return {"iu": iu, "mean_IU": mean_IU, "mean_IU_no_back": mean_IU_no_back, "mean_pixel_acc": mean_pixel_acc,
                "hist": hist, "labeled": labeled, "correct": correct}
