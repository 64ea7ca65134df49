"""`soil.util` helpers as far as the acceptance scripts call them
(python/soillib/util.py in the reference: relief shading and the matplotlib
viewers used by example/erosion_gpu.py:112 and example/tiff_relief.py).
Plain Python; matplotlib is imported lazily so that headless runs work."""
import numpy as np

from . import silt
from . import soil as _soil


def relief_shade(height, normal, light=(-1.0, 2.0, 1.0)):
    """Lambertian hill-shade of a height map with its normal map (util.py:75-100)."""
    light = np.asarray(light, np.float64)
    light = light / np.linalg.norm(light)
    diffuse = np.clip(np.sum(light * normal, axis=-1), 0.0, 1.0)
    h = np.asarray(height, np.float64)
    span = np.nanmax(h) - np.nanmin(h)
    flat = (h - np.nanmin(h)) / span if span > 0 else np.zeros_like(h)
    return 0.25 + 0.75 * diffuse * (0.5 + 0.5 * flat)


def show_relief(tensor, scale=(1.0, 1.0, 1.0), show=False):
    """util.py:119-125: normal map -> shaded relief -> imshow."""
    host = tensor.cpu() if tensor.host is silt.gpu else tensor
    normal = _soil.normal(host, scale).numpy()
    relief = relief_shade(host.numpy(), normal)
    import matplotlib.pyplot as plt
    plt.imshow(relief, cmap="gray")
    if show:
        plt.show()
    return relief


def show_height(tensor, show=False):
    host = tensor.cpu() if tensor.host is silt.gpu else tensor
    import matplotlib.pyplot as plt
    plt.imshow(host.numpy())
    if show:
        plt.show()


def show_discharge(tensor, show=False):
    host = tensor.cpu() if tensor.host is silt.gpu else tensor
    import matplotlib.pyplot as plt
    from matplotlib import colors
    data = np.maximum(host.numpy(), 1e-12)
    plt.imshow(data, cmap="CMRmap", norm=colors.LogNorm(data.min(), data.max()))
    if show:
        plt.show()
