"""`soil.util` helpers as far as the acceptance scripts call them
(python/soillib/util.py in the reference: relief shading and the matplotlib
viewers used by example/erosion_gpu.py:112 and example/tiff_relief.py).
Plain Python; matplotlib is imported lazily so that headless runs work."""
import numpy as np

from . import silt
from . import soil as _soil


def iter_tiff(source, max_files=None):
    """(name, path) pairs for `source`: the file itself, or the entries of a directory in sorted order,
    at most `max_files` of them.  BASELINE config 1's script walks its input with this
    (example/tiff_normal.py:9); raises RuntimeError for anything that is neither file nor directory."""
    from itertools import islice
    from pathlib import Path
    src = Path(source)
    if src.is_file():
        entries = [src]
    elif src.is_dir():
        entries = islice(sorted(e for e in src.iterdir() if e.is_file()), max_files)
    else:
        raise RuntimeError("iter_tiff: %r is neither a file nor a directory" % (str(source),))
    for entry in entries:
        yield entry.name, str(entry)


def relief_shade(height, normal):
    """Diffuse shade of a normal map under the light (-1, 2, 1) (util.py:32-52).  The
    height only enters the reference through terms whose weight is 0 there."""
    light = np.array([-1.0, 2.0, 1.0])
    light = light / np.linalg.norm(light)
    return np.sum(light * np.asarray(normal), axis=-1)


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
