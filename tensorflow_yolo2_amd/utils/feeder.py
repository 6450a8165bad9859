"""Host -> device input pipeline of a fed training loop (the `{input_data: image, label_data: gt_labels}` feed of
src/pascal/pascal_train_darknet.py:96-102, where TF copies 4 bytes per value synchronously every step).

Two pinned host slots and two device slots; batch i+1 is assembled and uploaded on its own HIP stream while step i
runs, as uint8 pixels (1 byte per value: 33 MB instead of 133 MB at 64 x 416 x 416 x 3; the float conversion
x / 255 * 2 - 1 runs inside the input pack kernel, y2_forward_u8).  PyTorch supplies pinned memory, streams and
events only."""
import numpy as np
import torch


class DeviceFeeder:
    def __init__(self, produce, batch, image_size, cell_size, label_depth=25, device="cuda:0", slots=2):
        """produce(images_u8 [B,size,size,3] uint8 numpy view, labels [B,S,S,depth] float32 numpy view) fills one batch"""
        self.produce = produce
        self.device = torch.device(device)
        self.slots = slots
        ishape, lshape = (batch, image_size, image_size, 3), (batch, cell_size, cell_size, label_depth)
        self.h_img = [torch.empty(ishape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.h_lab = [torch.empty(lshape, dtype=torch.float32).pin_memory() for _ in range(slots)]
        self.d_img = [torch.empty(ishape, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self.d_lab = [torch.empty(lshape, dtype=torch.float32, device=self.device) for _ in range(slots)]
        self.stream = torch.cuda.Stream(device=self.device)
        self.ready = [None] * slots        # upload of the slot has finished (recorded on the upload stream)
        self.free = [None] * slots         # the consumer has finished reading the slot (recorded on its stream)
        self.host_free = [None] * slots    # the upload has finished reading the slot's pinned memory
        self._next = 0
        self._queued = 0
        self.prefetch()

    def prefetch(self):
        """assemble the next batch on the host and queue its upload; returns at once (the copy is asynchronous)"""
        k = self._next
        if self.host_free[k] is not None:
            self.host_free[k].synchronize()         # the previous upload out of this pinned slot is done
        self.produce(self.h_img[k].numpy(), self.h_lab[k].numpy())
        with torch.cuda.stream(self.stream):
            if self.free[k] is not None:
                self.stream.wait_event(self.free[k])    # the step that read this device slot has finished with it
            self.d_img[k].copy_(self.h_img[k], non_blocking=True)
            self.d_lab[k].copy_(self.h_lab[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.ready[k] = ev
        self.host_free[k] = ev
        self._next = (k + 1) % self.slots
        self._queued += 1

    def get(self):
        """(images_u8, labels) device tensors of the oldest queued batch; the current stream waits for its upload.
        Call release() after queueing the work that reads them, then prefetch() for the batch after."""
        assert self._queued > 0, "prefetch() first"
        k = (self._next - self._queued) % self.slots
        torch.cuda.current_stream().wait_event(self.ready[k])
        self._cur = k
        return self.d_img[k], self.d_lab[k]

    def release(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free[self._cur] = ev
        self._queued -= 1
