"""ORACLE (test infrastructure only -- never imported by the product path).

Runs the REFERENCE'S OWN URDF reader and forward kinematics
(/root/reference/src/dex_retargeting/yourdfpy.py: ``URDF.load`` :896-959, ``_parse_origin`` :1375-1387,
``_parse_axis`` :1631-1643, ``_parse_mimic`` :1107-1115, ``_add_dummy_joints`` :1942-1984,
``_forward_kinematics_joint`` :1013-1050, ``build_tree`` / ``update_kinematics`` / ``get_link_global_transform``
:1862-1939, ``write_xml_file`` :1098-1105), imported from where it lies -- nothing is copied -- with stand-ins for the
third-party packages it imports that are not installed in this container:

* ``lxml.etree``      -> the standard library's ``xml.etree.ElementTree`` (same element API for what yourdfpy uses);
* ``anytree``         -> a 30-line ``Node`` / ``LevelOrderIter`` / ``search.findall_by_attr``;
* ``pytransform3d``   -> ``rotations.matrix_from_euler / euler_from_matrix / matrix_from_axis_angle /
  matrix_from_quaternion`` and ``transformations.transform_from``, restated from the published definitions
  [not-in-ref: third-party conventions -- active rotation matrices, extrinsic = rotations about fixed axes applied in
  the order i, j, k, i.e. R = R_k(e2) R_j(e1) R_i(e0)].

What this pins (tests/golden/fk_golden.npz, written by tests/golden/gen_golden.py): the URDF conventions
(rpy order, default axis, non-unit axes, mimic parameters), fixed-joint handling, dummy free joints, and the
composition of the chain -- as the reference's own code computes them -- for every fixture URDF with and without
dummy joints.  What it cannot pin: pinocchio's dof ORDER (the reference's reader keeps URDF file order); all
comparisons are therefore keyed by joint and link NAME.

Only usable where /root/reference exists (this build container).
"""
from __future__ import annotations

import sys
import types
import xml.etree.ElementTree as ET
from collections import deque

import numpy as np

from .ref_harness import REFERENCE_SRC, reference_available


# ---- pytransform3d stand-ins ------------------------------------------------------------------------------
def _basis_rotation(axis: int, angle: float) -> np.ndarray:
    """Active rotation about basis vector `axis` (0 = x, 1 = y, 2 = z)."""
    c, s = np.cos(angle), np.sin(angle)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def matrix_from_euler(e, i, j, k, extrinsic):
    a, b, c = (float(v) for v in e)
    if not extrinsic:  # intrinsic i-j'-k'' == extrinsic k-j-i with the angles swapped
        i, k = k, i
        a, c = c, a
    return _basis_rotation(k, c) @ _basis_rotation(j, b) @ _basis_rotation(i, a)


def euler_from_matrix(R, i, j, k, extrinsic):
    """Inverse of matrix_from_euler for the proper Tait-Bryan sequence (0, 1, 2) -- the only one the reference uses
    (yourdfpy.py:1401, seq_retarget.py:96-98)."""
    R = np.asarray(R, dtype=np.float64)
    assert (i, j, k) == (0, 1, 2)
    if extrinsic:  # R = Rz(c) Ry(b) Rx(a)
        b = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
        a = np.arctan2(R[2, 1], R[2, 2])
        c = np.arctan2(R[1, 0], R[0, 0])
    else:  # R = Rx(a) Ry(b) Rz(c)
        b = np.arcsin(np.clip(R[0, 2], -1.0, 1.0))
        a = np.arctan2(-R[1, 2], R[2, 2])
        c = np.arctan2(-R[0, 1], R[0, 0])
    return np.array([a, b, c])


def matrix_from_axis_angle(a):
    """(ux, uy, uz, theta) -> rotation matrix (Rodrigues), axis used as given (pytransform3d does not re-normalise)."""
    ux, uy, uz, th = (float(v) for v in a)
    c, s = np.cos(th), np.sin(th)
    ci = 1.0 - c
    return np.array([[ci * ux * ux + c, ci * ux * uy - uz * s, ci * ux * uz + uy * s],
                     [ci * uy * ux + uz * s, ci * uy * uy + c, ci * uy * uz - ux * s],
                     [ci * uz * ux - uy * s, ci * uz * uy + ux * s, ci * uz * uz + c]])


def matrix_from_quaternion(q):  # (w, x, y, z)
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def transform_from(R, p):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p
    return T


# ---- anytree stand-in ----------------------------------------------------------------------------------------
class Node:
    def __init__(self, name, parent=None, **kw):
        self.name = name
        self.parent = parent
        self.children = []
        if parent is not None:
            parent.children.append(self)
        for k, v in kw.items():
            setattr(self, k, v)


def LevelOrderIter(root):
    todo = deque([root])
    while todo:
        n = todo.popleft()
        yield n
        todo.extend(n.children)


def _findall_by_attr(node, value, name="name"):
    return tuple(n for n in LevelOrderIter(node) if getattr(n, name, None) == value)


class _ElementTree(ET.ElementTree):
    def write(self, fname, xml_declaration=True, pretty_print=False, **kw):  # lxml's signature
        if pretty_print:
            ET.indent(self, space="  ")
        super().write(fname, xml_declaration=xml_declaration, encoding="utf-8")


def install_stubs():
    if "pytransform3d" not in sys.modules or not hasattr(sys.modules["pytransform3d"], "transformations"):
        pt = sys.modules.get("pytransform3d") or types.ModuleType("pytransform3d")
        rot = types.ModuleType("pytransform3d.rotations")
        rot.matrix_from_euler = matrix_from_euler
        rot.euler_from_matrix = euler_from_matrix
        rot.matrix_from_axis_angle = matrix_from_axis_angle
        rot.matrix_from_quaternion = matrix_from_quaternion
        tr = types.ModuleType("pytransform3d.transformations")
        tr.transform_from = transform_from
        pt.rotations, pt.transformations = rot, tr
        sys.modules["pytransform3d"] = pt
        sys.modules["pytransform3d.rotations"] = rot
        sys.modules["pytransform3d.transformations"] = tr
    if "anytree" not in sys.modules:
        at = types.ModuleType("anytree")
        at.Node, at.LevelOrderIter = Node, LevelOrderIter
        at.search = types.ModuleType("anytree.search")
        at.search.findall_by_attr = _findall_by_attr
        sys.modules["anytree"] = at
        sys.modules["anytree.search"] = at.search
    if "lxml" not in sys.modules:
        lx = types.ModuleType("lxml")
        et = types.ModuleType("lxml.etree")
        et.XMLParser = lambda **kw: ET.XMLParser()  # xml.etree drops comments by default
        et.parse = lambda f, parser=None: ET.parse(f, parser=parser)
        et.strip_tags = lambda *a, **k: None
        et.cleanup_namespaces = lambda *a, **k: None
        et.Comment = ET.Comment
        et.Element, et.SubElement, et.ElementTree = ET.Element, ET.SubElement, _ElementTree
        et.tostring = lambda e, xml_declaration=True, **kw: ET.tostring(e.getroot() if hasattr(e, "getroot") else e)
        lx.etree = et
        sys.modules["lxml"] = lx
        sys.modules["lxml.etree"] = et
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def import_reference_yourdfpy():
    if not reference_available():
        raise RuntimeError("/root/reference is not present on this machine")
    install_stubs()
    import dex_retargeting.yourdfpy as y  # noqa: E402

    return y


class _HomogeneousRotations:
    """``_forward_kinematics_joint`` multiplies the 4x4 joint origin by ``rotations.matrix_from_axis_angle(...)``
    (yourdfpy.py:1044-1046), which is 3x3 in pytransform3d: as written the reference's FK raises for every revolute
    joint (upstream yourdfpy used trimesh's 4x4 ``rotation_matrix`` there; the vendored copy swapped the dependency).
    To execute the reference's tree code (build_tree, update_kinematics, mimic handling, level-order composition) the
    module-level name ``rotations`` is replaced, inside the imported yourdfpy module only, by this proxy whose
    ``matrix_from_axis_angle`` embeds the same rotation in a homogeneous matrix -- the evident intent of the line."""

    def __getattr__(self, name):
        return getattr(sys.modules["pytransform3d.rotations"], name)

    @staticmethod
    def matrix_from_axis_angle(a):
        return transform_from(matrix_from_axis_angle(a), np.zeros(3))


def load_reference_urdf(path: str, add_dummy_free_joints: bool):
    """URDF.load as RetargetingConfig.build calls it (retargeting_config.py:176-180) + the kinematic tree."""
    y = import_reference_yourdfpy()
    if not isinstance(y.rotations, _HomogeneousRotations):
        y.rotations = _HomogeneousRotations()
    u = y.URDF.load(path, add_dummy_free_joints=add_dummy_free_joints, build_scene_graph=False)
    # build_scene_graph=False leaves the base link undetermined (yourdfpy.py:611-614); build_tree needs it
    u._base_link = u._determine_base_link()
    u.tree_root = u.build_tree()
    return u


def reference_link_transforms(u, cfg: np.ndarray) -> np.ndarray:
    """cfg over u.actuated_joint_names -> (n_links, 4, 4) global transforms in u.robot.links order.  Mimic joints read
    the model's stored configuration (yourdfpy.py:1017-1023), so it is set first."""
    u._cfg = np.asarray(cfg, dtype=np.float64)
    u.update_kinematics(np.asarray(cfg, dtype=np.float64))
    return np.stack([u.get_link_global_transform(l.name) for l in u.robot.links])
