"""FFT operations over hipFFT against numpy.fft (reference test/test_fft.py)."""

import numpy as np
import pytest

from katsdpsigproc_amd import accel


def test_argument_checks_need_no_gpu():
    from katsdpsigproc_amd import fft

    with pytest.raises(TypeError):  # not a HIP context
        fft.FftTemplate(object(), 1, (8,), np.complex64, np.complex64, (8,), (8,))
    assert fft.FftMode.FORWARD.value == 0 and fft.FftMode.INVERSE.value == 1


@pytest.fixture(scope="module")
def context():
    return accel.create_some_context(interactive=False)


@pytest.fixture(scope="module")
def command_queue(context):
    return context.create_command_queue()


@pytest.mark.gpu
class TestFft:
    def _run(self, context, command_queue, template, mode, src):
        from katsdpsigproc_amd import fft

        fn = template.instantiate(command_queue, mode)
        fn.ensure_all_bound()
        fn.buffer("src").set(command_queue, src)
        fn()
        assert ("work_area" in fn.slots) == (template._work_size > 0)
        return fn.buffer("dest").get(command_queue)

    @pytest.mark.parametrize("precision", [np.complex64, np.complex128])
    def test_complex_2d_batched_with_padding(self, precision, context, command_queue):
        from katsdpsigproc_amd import fft

        shape, padded_src, padded_dest = (3, 2, 48, 64), (3, 2, 50, 72), (3, 2, 48, 80)
        rs = np.random.RandomState(1)
        src = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(precision)
        template = fft.FftTemplate(context, 2, shape, precision, precision, padded_src, padded_dest)
        tol = 1e-4 if precision == np.complex64 else 1e-11
        out = self._run(context, command_queue, template, fft.FftMode.FORWARD, src)
        np.testing.assert_allclose(out, np.fft.fftn(src, axes=(2, 3)), rtol=tol, atol=tol * 50)
        out = self._run(context, command_queue, template, fft.FftMode.INVERSE, src)
        expected = np.fft.ifftn(src, axes=(2, 3)) * (48 * 64)  # unnormalised
        np.testing.assert_allclose(out, expected, rtol=tol, atol=tol * 50)

    def test_real_transforms(self, context, command_queue):
        from katsdpsigproc_amd import fft

        shape = (5, 30, 40)
        rs = np.random.RandomState(2)
        real = rs.standard_normal(shape).astype(np.float32)
        r2c = fft.FftTemplate(context, 2, shape, np.float32, np.complex64, (5, 30, 48), (5, 30, 24))
        spectrum = self._run(context, command_queue, r2c, fft.FftMode.FORWARD, real)
        assert spectrum.shape == (5, 30, 21)
        np.testing.assert_allclose(spectrum, np.fft.rfftn(real, axes=(1, 2)), rtol=1e-4, atol=5e-3)
        c2r = fft.FftTemplate(context, 2, shape, np.complex64, np.float32, (5, 30, 21), (5, 30, 40))
        back = self._run(context, command_queue, c2r, fft.FftMode.INVERSE, spectrum)
        np.testing.assert_allclose(back, real * (30 * 40), rtol=1e-4, atol=5e-2)
        with pytest.raises(ValueError):
            r2c.instantiate(command_queue, fft.FftMode.INVERSE)
        with pytest.raises(ValueError):
            c2r.instantiate(command_queue, fft.FftMode.FORWARD)

    def test_bad_shapes(self, context):
        from katsdpsigproc_amd import fft

        with pytest.raises(ValueError):  # batch dimension padded
            fft.FftTemplate(context, 1, (4, 16), np.complex64, np.complex64, (5, 16), (4, 16))
        with pytest.raises(ValueError):
            fft.FftTemplate(context, 1, (4, 16), np.complex64, np.complex64, (4, 16), (16,))
        with pytest.raises(ValueError):
            fft.FftTemplate(context, 1, (4, 16), np.float32, np.float32, (4, 16), (4, 16))
