#!/usr/bin/env python3
"""Generate include/go2_model_data.h (a table of numbers) from the Go2 URDF.

Container-only tool: reads /root/reference/resources/robots/go2/urdf/go2.urdf (the robot
description the reference loads at legged_gym/envs/base/legged_robot.py:961-980 with the asset
options of legged_gym/envs/base/legged_robot_config.py:114-134) and emits the constants the
oracle and the HIP kernels need:

  * the 19 rigid bodies left after `collapse_fixed_joints=True` (fixed joints are merged into the
    parent except those tagged dont_collapse="true", URDF :72,110,369,628,887,1146),
  * for each body: the moving link it is rigidly attached to (0 = base, 1+3*leg+j = hip/thigh/calf),
    its offset in that link's frame, mass, COM and inertia (about the COM, body axes),
  * the 12 revolute joints in DOF order FL,FR,RL,RR x hip,thigh,calf (go2_env.py:56-58),
  * collision candidates: every collision primitive reduced to spheres (sphere -> itself,
    cylinder -> capsule end spheres because replace_cylinder_with_capsule=True, box -> 8 corners).

The output holds only numbers derived from the URDF; no reference source text is copied.
"""
import sys, math
import xml.etree.ElementTree as ET
import numpy as np

URDF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/robots/go2/urdf/go2.urdf"
OUT = sys.argv[2] if len(sys.argv) > 2 else "include/go2_model_data.h"

def vec(s): return np.array([float(x) for x in s.split()])
def rpy_mat(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1,0,0],[0,cr,-sr],[0,sr,cr]]); Ry = np.array([[cp,0,sp],[0,1,0],[-sp,0,cp]])
    Rz = np.array([[cy,-sy,0],[sy,cy,0],[0,0,1]])
    return Rz @ Ry @ Rx
def origin(e):
    o = e.find("origin")
    if o is None: return np.eye(3), np.zeros(3)
    return rpy_mat(*vec(o.get("rpy", "0 0 0"))), vec(o.get("xyz", "0 0 0"))

root = ET.parse(URDF).getroot()
links = {l.get("name"): l for l in root.findall("link")}
link_order = [l.get("name") for l in root.findall("link")]
joints = root.findall("joint")
child_joint = {j.find("child").get("link"): j for j in joints}

LEGS = ["FL", "FR", "RL", "RR"]
DOF_NAMES = [f"{l}_{j}_joint" for l in LEGS for j in ("hip", "thigh", "calf")]
MOVING = ["base"] + [f"{l}_{j}" for l in LEGS for j in ("hip", "thigh", "calf")]

def chain_to_moving(name):
    """Return (moving link index, R, t) of link `name` frame expressed in its moving ancestor."""
    R, t = np.eye(3), np.zeros(3)
    cur = name
    while cur not in MOVING:
        j = child_joint[cur]
        assert j.get("type") == "fixed", (cur, j.get("type"))
        Rj, tj = origin(j)
        R, t = Rj @ R, Rj @ t + tj
        cur = j.find("parent").get("link")
    return MOVING.index(cur), R, t

# ---- bodies that survive collapsing ------------------------------------------------------------
def survives(name):
    if name in MOVING: return True
    j = child_joint[name]
    return j.get("dont_collapse", "false") == "true"
BODY_NAMES = [n for n in link_order if survives(n)]
# order used throughout the build: base, heads, then per leg hip,thigh,calf,foot
want = ["base", "Head_upper", "Head_lower"] + [f"{l}_{p}" for l in LEGS for p in ("hip", "thigh", "calf", "foot")]
assert sorted(want) == sorted(BODY_NAMES), BODY_NAMES
BODY_NAMES = want

def owner_body(name):
    cur = name
    while cur not in BODY_NAMES:
        cur = child_joint[cur].find("parent").get("link")
    return BODY_NAMES.index(cur)

bodies = []
for b in BODY_NAMES:
    mi, R, t = chain_to_moving(b)
    assert np.allclose(R, np.eye(3)), "body frames are axis-aligned with their moving link in this URDF"
    m, c, I = 0.0, np.zeros(3), np.zeros((3, 3))
    # merge inertials of all collapsed descendants (only zero-mass links collapse in this URDF)
    for ln in link_order:
        if owner_body(ln) != BODY_NAMES.index(b): continue
        ine = links[ln].find("inertial")
        if ine is None: continue
        mm = float(ine.find("mass").get("value"))
        if mm == 0.0: continue
        assert ln == b, "only the body's own link carries mass in this URDF"
        Ri, ci = origin(ine)
        assert np.allclose(Ri, np.eye(3))
        it = ine.find("inertia")
        Ii = np.array([[float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                       [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                       [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))]])
        m, c, I = mm, ci, Ii
    bodies.append(dict(name=b, link=mi, off=t, mass=m, com=c, I=I))
total_mass = sum(b["mass"] for b in bodies)
assert abs(total_mass - 15.019) < 1e-9, total_mass

# ---- joints -------------------------------------------------------------------------------------
jt = []
for n in DOF_NAMES:
    j = [x for x in joints if x.get("name") == n][0]
    R, t = origin(j); assert np.allclose(R, np.eye(3))
    lim = j.find("limit")
    parent = MOVING.index(j.find("parent").get("link")); child = MOVING.index(j.find("child").get("link"))
    jt.append(dict(name=n, parent=parent, child=child, xyz=t, axis=vec(j.find("axis").get("xyz")),
                   lower=float(lim.get("lower")), upper=float(lim.get("upper")),
                   effort=float(lim.get("effort")), velocity=float(lim.get("velocity"))))
    assert child == len(jt)

# ---- collision candidates -----------------------------------------------------------------------
spheres = []  # (body, link, center(link frame), radius, kind)
for ln in link_order:
    for col in links[ln].findall("collision"):
        mi, R, t = chain_to_moving(ln)
        Rc, tc = origin(col)
        Rw, tw = R @ Rc, R @ tc + t
        g = col.find("geometry")[0]
        body = owner_body(ln)
        if g.tag == "sphere":
            spheres.append((body, mi, tw, float(g.get("radius")), 0))
        elif g.tag == "cylinder":
            r, L = float(g.get("radius")), float(g.get("length"))
            ax = Rw @ np.array([0, 0, 1.0])
            spheres.append((body, mi, tw + 0.5 * L * ax, r, 1))
            spheres.append((body, mi, tw - 0.5 * L * ax, r, 1))
            if mi != 0 and L >= 0.06:
                # replace_cylinder_with_capsule (legged_robot.py:965-980): a capsule is its two end spheres EXACTLY on a plane (the lowest point of a
                # segment is an end); on the trimesh a stair nosing can enter BETWEEN the ends of a long one.  Round 4: the two long calf capsules
                # (0.12 m, 0.065 m) carry a mid-segment sphere (kind 4: sorted behind the link's end spheres below, so ties on a plane go to an end)
                spheres.append((body, mi, tw, r, 4))
        elif g.tag == "box":
            sx, sy, sz = vec(g.get("size"))
            for a in (-1, 1):
                for b_ in (-1, 1):
                    for c_ in (-1, 1):
                        spheres.append((body, mi, tw + Rw @ np.array([a * sx / 2, b_ * sy / 2, c_ * sz / 2]), 0.0, 2))
            if mi != 0:
                # the thigh box (0.11 m long, go2.urdf:206-209): midpoints of its four long edges (kind 4), for the same reason
                ext = [sx, sy, sz]; la = int(np.argmax(ext))
                for b_ in (-1, 1):
                    for c_ in (-1, 1):
                        v = [0.0, 0.0, 0.0]; oth = [k for k in range(3) if k != la]
                        v[oth[0]] = b_ * ext[oth[0]] / 2; v[oth[1]] = c_ * ext[oth[1]] / 2
                        spheres.append((body, mi, tw + Rw @ np.array(v), 0.0, 4))
            if mi == 0 and BODY_NAMES[body] == "base":
                # The trunk box is the one primitive long enough (0.376 m) for the ground to reach it BETWEEN its corners — a stair nosing or
                # an obstacle edge under the belly (mesh_type 'trimesh').  Extra surface samples, appended after all other base points below
                # (kind 3): the bottom face centre and the midpoints of the four long edges.  On a plane a box's lowest point is a corner, so
                # nothing changes there (ties go to the corner, which comes first in the scan order).
                for a, b_, c_ in ((0, 0, -1), (0, -1, -1), (0, 1, -1), (0, -1, 1), (0, 1, 1)):
                    spheres.append((body, mi, tw + Rw @ np.array([a * sx / 2, b_ * sy / 2, c_ * sz / 2]), 0.0, 3))
        else:
            raise ValueError(g.tag)
# group: feet first (one per leg), then per-leg non-foot, then base-attached
feet = [s for s in spheres if BODY_NAMES[s[0]].endswith("foot")]
assert len(feet) == 4
# per leg: link by link (hip, thigh, calf), inside a link the primitive's own points first and the round-4 flank samples (kind 4) behind them
leg_other = [[s for lk in (1, 2, 3) for kinds in ((0, 1, 2), (4,)) for s in spheres
              if s[1] == lk + 3 * l and s[4] in kinds and not BODY_NAMES[s[0]].endswith("foot")] for l in range(4)]
base_pts = [s for s in spheres if s[1] == 0 and s[4] != 3] + [s for s in spheres if s[1] == 0 and s[4] == 3]
assert len(base_pts) <= 16, "the lane programs deal the base points to 4 legs x 4 sub-lanes (go2_tables.h)"
n_leg_other = len(leg_other[0]); assert all(len(x) == n_leg_other for x in leg_other)

def f(x): return repr(float(np.float64(x))) if abs(x) > 0 else "0.0"
def arr(v): return "{" + ", ".join(f(x) for x in np.asarray(v).ravel()) + "}"

o = []
o.append("/* GENERATED by tools/gen_go2_model.py from the Go2 URDF (reference: resources/robots/go2/urdf/go2.urdf,")
o.append(" * loaded at legged_gym/envs/base/legged_robot.py:961-980). Numbers only. Do not edit. */")
o.append("#ifndef GO2_MODEL_DATA_H\n#define GO2_MODEL_DATA_H\n")
o.append(f"#define GO2_NUM_BODIES {len(bodies)}\n#define GO2_NUM_DOF 12\n#define GO2_NUM_LEGS 4\n#define GO2_NUM_LINKS 13")
o.append(f"#define GO2_LEG_OTHER_PTS {n_leg_other}   /* non-foot collision candidates per leg */")
for nm, lk in (("HIP", 1), ("THIGH", 2), ("CALF", 3)):
    o.append(f"#define GO2_N_{nm}_PTS {sum(1 for s in leg_other[0] if s[1] == lk)}")
o.append(f"#define GO2_BASE_PTS {len(base_pts)}       /* candidates rigidly attached to the base (base box, heads) */")
o.append(f"#define GO2_TOTAL_MASS {f(total_mass)}\n")
o.append("/* body order used for contact_forces[N,19,3] / rigid_body_states[N,19,13] */")
o.append("#define GO2_BODY_NAMES_INIT {" + ", ".join('"%s"' % b["name"] for b in bodies) + "}")
o.append("#define GO2_DOF_NAMES_INIT {" + ", ".join('"%s"' % n for n in DOF_NAMES) + "}")
o.append("/* moving link each body is welded to: 0 = base, 1+3*leg+j (j=0 hip,1 thigh,2 calf) */")
o.append("#define GO2_BODY_LINK_INIT {" + ", ".join(str(b["link"]) for b in bodies) + "}")
o.append("/* body frame origin in its moving-link frame [19][3] */")
o.append("#define GO2_BODY_OFFSET_INIT {" + ", ".join(arr(b["off"]) for b in bodies) + "}")
o.append("#define GO2_BODY_MASS_INIT {" + ", ".join(f(b["mass"]) for b in bodies) + "}")
o.append("/* COM in body frame [19][3] */")
o.append("#define GO2_BODY_COM_INIT {" + ", ".join(arr(b["com"]) for b in bodies) + "}")
o.append("/* rotational inertia about the COM, body axes: xx,yy,zz,xy,xz,yz [19][6] */")
o.append("#define GO2_BODY_INERTIA_INIT {" + ", ".join(arr([b["I"][0,0], b["I"][1,1], b["I"][2,2], b["I"][0,1], b["I"][0,2], b["I"][1,2]]) for b in bodies) + "}")
o.append("\n/* joints in DOF order; joint i moves link i+1 */")
o.append("#define GO2_JOINT_PARENT_INIT {" + ", ".join(str(j["parent"]) for j in jt) + "}")
o.append("#define GO2_JOINT_ORIGIN_INIT {" + ", ".join(arr(j["xyz"]) for j in jt) + "}")
o.append("/* axis: 0 = x, 1 = y (all Go2 joints are axis aligned) */")
for j in jt: assert list(j["axis"]) in ([1,0,0],[0,1,0])
o.append("#define GO2_JOINT_AXIS_INIT {" + ", ".join(str(int(np.argmax(j["axis"]))) for j in jt) + "}")
o.append("#define GO2_JOINT_LOWER_INIT {" + ", ".join(f(j["lower"]) for j in jt) + "}")
o.append("#define GO2_JOINT_UPPER_INIT {" + ", ".join(f(j["upper"]) for j in jt) + "}")
o.append("#define GO2_JOINT_EFFORT_INIT {" + ", ".join(f(j["effort"]) for j in jt) + "}")
o.append("#define GO2_JOINT_VELOCITY_INIT {" + ", ".join(f(j["velocity"]) for j in jt) + "}")
o.append("\n/* collision candidates as spheres: {body, link, cx, cy, cz (link frame), radius} */")
def sph(s): return "{%d, %d, %s, %s, %s, %s}" % (s[0], s[1], f(s[2][0]), f(s[2][1]), f(s[2][2]), f(s[3]))
o.append("#define GO2_FOOT_PTS_INIT {" + ", ".join(sph(s) for s in feet) + "}")
o.append("#define GO2_LEG_OTHER_PTS_INIT {" + ", \\\n  ".join("{" + ", ".join(sph(s) for s in lo) + "}" for lo in leg_other) + "}")
o.append("#define GO2_BASE_PTS_INIT {" + ", ".join(sph(s) for s in base_pts) + "}")
o.append("\n#endif")
open(OUT, "w").write("\n".join(o) + "\n")
print("wrote", OUT, "bodies", len(bodies), "leg_other", n_leg_other, "base_pts", len(base_pts), "mass", total_mass)
