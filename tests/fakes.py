"""Stand-ins for the HIP backend so that slot wiring and host logic can be tested
without a GPU: buffers are numpy arrays, kernels record their launches."""

import numpy as np

from katsdpsigproc_amd import accel


class FakeBuffer:
    def __init__(self, shape, dtype):
        self.array = np.zeros(shape, dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = self.array.nbytes
        self.ptr = self.array.ctypes.data


class FakeKernel:
    def __init__(self, name):
        self.name = name


class FakeDevice:
    name = "fake"
    platform_name = "fake platform"
    driver_version = "0"
    simd_group_size = 64


class FakeContext:
    def __init__(self):
        self.device = FakeDevice()
        self.raw_allocations = []

    def native_kernel(self, name):
        return FakeKernel(name)

    def allocate_raw(self, n_bytes):
        raw = np.zeros(n_bytes, np.uint8)
        self.raw_allocations.append(raw)
        return raw

    def allocate(self, shape, dtype, raw=None):
        return FakeBuffer(shape, dtype)

    def allocate_pinned(self, shape, dtype):
        return np.empty(shape, dtype)

    def create_command_queue(self, profile=False):
        return FakeQueue(self)


class FakeQueue:
    def __init__(self, context):
        self.context = context
        self.launches = []

    def enqueue_kernel(self, kernel, args, global_size=None, local_size=None):
        self.launches.append((kernel.name, list(args)))

    def enqueue_write_buffer(self, buffer, data, blocking=True):
        buffer.array[...] = data

    def enqueue_read_buffer(self, buffer, data, blocking=True):
        data[...] = buffer.array

    def enqueue_zero_buffer(self, buffer):
        buffer.array[...] = 0

    def finish(self):
        pass


def make_queue():
    return FakeContext().create_command_queue()
