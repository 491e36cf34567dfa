"""Import the REFERENCE's Python package ``envs`` (from /root/reference) in this container.

Only usable here (the reference tree does not exist on the GPU box); used by the golden-vector
generators in this directory.  The reference imports ROS / OpenCV / gym, none of which this image
has.  None of them takes part in the arithmetic being pinned, so they are replaced by inert
import-level stand-ins:

  * message / request classes (comn_pkg, geometry_msgs, ...): attribute bags,
  * gym.Env / gym.Wrapper / gym.ObservationWrapper: the standard delegation only,
  * rospy services: a callable supplied by the generator (it plays the C++ node's part),
  * cv_bridge.imgmsg_to_cv2: pass-through; cv2.resize: identity, legal only for equal sizes
    (OpenCV's resize copies when dsize == src size),
  * tf.transformations.quaternion_from_euler: the published planar formula.

Everything under test -- ImageEnv._get_states / _draw_ped_map / step, the wrapper stack,
EnvPos -- is the reference's own code, unmodified, executed by the real interpreter.
"""
import importlib.abc
import importlib.machinery
import math
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class Msg:
    """permissive ROS message stand-in: attributes spring into existence as nested messages"""

    def __init__(self, *args, **kw):
        if args and self.__class__.__name__ == "Point":
            kw = dict(zip(("x", "y", "z"), args), **kw)
        self.__dict__.update(kw)

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = Msg()
        object.__setattr__(self, k, v)
        return v


def _msg_class(name, lists=(), defaults=None):
    def __init__(self, *a, **kw):
        Msg.__init__(self, *a, **kw)
        for l in lists:
            self.__dict__.setdefault(l, [])
        for k, v in (defaults or {}).items():
            self.__dict__.setdefault(k, v)
    return type(name, (Msg,), {"__init__": __init__})


class _Env:
    metadata = {}

    def reset(self, **kw):
        raise NotImplementedError

    def step(self, a):
        raise NotImplementedError


class _Wrapper(_Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        return getattr(self.env, k)

    def step(self, a):
        return self.env.step(a)

    def reset(self, **kw):
        return self.env.reset(**kw)


class _ObservationWrapper(_Wrapper):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, a):
        o, r, d, i = self.env.step(a)
        return self.observation(o), r, d, i


class ServiceException(Exception):
    pass


SERVICES = {}  # service name -> callable(request) -> response, filled by the generator


class _Permissive(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _msg_class(k)
        setattr(self, k, v)
        return v


_STUB_ROOTS = ("rospy", "rospkg", "rosbag", "imageio", "cv2", "cv_bridge", "tf", "tf2_ros", "tf2_msgs", "actionlib",
               "gym", "comn_pkg", "geometry_msgs", "nav_msgs", "std_msgs", "std_srvs", "sensor_msgs", "gazebo_msgs",
               "matplotlib", "pandas", "psutil")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Permissive(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        n = m.__name__
        if n == "gym":
            m.Env, m.Wrapper, m.ObservationWrapper = _Env, _Wrapper, _ObservationWrapper
        elif n == "rospy":
            m.init_node = lambda *a, **k: None
            m.wait_for_service = lambda *a, **k: None
            m.ServiceProxy = lambda name, typ: SERVICES[name.split("/")[-1]]
        elif n == "rospy.service":
            m.ServiceException = ServiceException
        elif n == "rospkg":
            m.RosPack = lambda: types.SimpleNamespace(get_path=lambda pkg: REFERENCE_ROOT + "/src/" + pkg)
        elif n == "cv_bridge":
            m.CvBridge = lambda: types.SimpleNamespace(imgmsg_to_cv2=lambda img, desired_encoding=None: img)
        elif n == "cv2":
            m.INTER_CUBIC = 2

            def resize(img, dsize, interpolation=None):
                assert (img.shape[1], img.shape[0]) == tuple(dsize), "identity resize only"
                return img.copy()
            m.resize = resize
        elif n == "tf":
            m.transformations = types.SimpleNamespace(
                quaternion_from_euler=lambda r, p, y: np.array([0.0, 0.0, math.sin(y / 2.0), math.cos(y / 2.0)]))
        elif n == "comn_pkg.msg":
            m.Agent = _msg_class("Agent", lists=("trajectory", "trajectory_v", "size", "sensor_cfg"))
            m.Env = _msg_class("Env", lists=("robots", "peds", "obstacles"))
        elif n == "comn_pkg.srv":
            m.StepEnvRequest = _msg_class("StepEnvRequest", lists=("robots",))
            m.ResetEnvRequest = _msg_class("ResetEnvRequest", lists=("robots", "peds", "obstacles"))
            m.EndEpRequest = _msg_class("EndEpRequest", lists=("robot_res",))
        elif n == "geometry_msgs.msg":
            m.Point = _msg_class("Point")


_installed = False


def import_reference_envs():
    """returns the reference's ``envs`` package"""
    global _installed
    import torch  # noqa: F401  (the reference imports torch; load the real one before the finder)
    import scipy  # noqa: F401
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        sys.path.insert(0, REFERENCE_ROOT)
        _installed = True
    import envs
    return envs
