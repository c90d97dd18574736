"""The wide one-launch step on a degenerate table: all-zero rows, all-zero codebook (every node ties: the first must win)."""
import numpy as np
import torch
from ark_analysis_amd import som_device
from ark_analysis_amd.schedule import BatchSchedule
from ark_analysis_amd.flowsom import default_radius_range

dev = torch.device("cuda", 0)
for dtype in (torch.float64, torch.float32):
    for n, c, pad, zero_w in ((2, 2, 7, True), (2, 2, 7, False), (300, 2, 7, True), (2, 5, 0, True)):
        buf = torch.zeros((n, c + pad), dtype=dtype)
        x = buf.to(dev)[:, :c]
        k = 100
        w0 = np.zeros((k, c)) if zero_w else np.random.RandomState(1).rand(k, c)
        sch = BatchSchedule.equal(16)
        for unfused in (False, True):
            st = som_device.BatchTrainState(n, c, 10, 10, sch, dev, dtype=dtype)
            st.wbuf[0].copy_(torch.from_numpy(w0))
            som_device.batch_train_steps(x, st, 0, 1, 16, (0.05, 0.01), default_radius_range(10, 10), unfused=unfused)
            ring = st.ring[0].cpu().numpy()
            cnt = ring[k * c:]
            print(dtype, "n", n, "c", c, "pad", pad, "zero_w", zero_w, "unfused", unfused, "counts at", np.nonzero(cnt)[0], cnt[np.nonzero(cnt)[0]])
