"""Inert stand-ins for the third-party packages the reference's Nerfstudio tree imports and this image lacks (tyro, viser,
torchmetrics, cv2, ...), so that the tree's OWN host classes -- MethodSpecification, TrainerConfig, Model, Field, Pipeline,
the registry -- can be imported in the build container and the plugin surface checked against them
(tests/test_api_surface.py::test_reference_registry_accepts_the_spec).  Test infrastructure; takes part in no arithmetic.
A stand-in module answers every attribute with a class that can be subscripted, called, used as a decorator or as a base."""
import importlib.abc
import importlib.machinery
import sys
import types

ROOTS = {"tyro", "imageio", "cv2", "torchvision", "wandb", "nerfacc", "torchtyping", "jaxtyping", "viser", "mediapy", "plotly", "torchmetrics",
         "PIL", "tensorboard", "open3d", "pymeshlab", "xatlas", "pyquaternion", "msgpack_numpy", "umsgpack", "zmq", "tornado", "aiortc", "av",
         "appdirs", "cryptography", "pyngrok", "nuscenes", "lpips", "skimage", "h5py", "gdown", "functorch", "splines", "trimesh", "websockets",
         "pytorch_msssim", "comet_ml", "matplotlib", "tinycudann", "open_clip", "diffusers", "rawpy", "pathos", "kornia", "msgpack",
         "importlib_metadata", "torch_scatter", "nvdiffrast", "awscli", "sshconf", "jupyterlab", "ipywidgets"}


class _Meta(type):
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any

    def __getitem__(cls, i):
        return cls

    def __call__(cls, *a, **kw):
        return type.__call__(cls)

    def __or__(cls, o):
        return cls

    def __ror__(cls, o):
        return cls


class _Any(metaclass=_Meta):
    def __init__(self, *a, **kw):
        pass

    def __call__(self, *a, **kw):
        if len(a) == 1 and callable(a[0]) and not kw:  # used as a decorator: hand the function back
            return a[0]
        return _Any()

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()

    def __getitem__(self, i):
        return _Any()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _Module(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k.endswith("Warning"):
            return type(k, (Warning,), {})
        return _Any


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in ROOTS:
            try:  # a package the image does have is left alone
                if name.split(".")[0] in _real:
                    return None
            except Exception:  # noqa: BLE001
                pass
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Module(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


_real = set()


def install():
    import importlib.util

    for r in list(ROOTS):
        try:
            if importlib.util.find_spec(r) is not None:
                _real.add(r)
        except (ImportError, ValueError):
            pass
    sys.meta_path.insert(0, _Finder())
    sys.modules["torch.utils.tensorboard"] = _Module("torch.utils.tensorboard")
