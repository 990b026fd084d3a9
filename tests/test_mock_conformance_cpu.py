"""The adapter (mov-slam_amd/host/Optimizer.cc) has only ever been compiled against the mock map classes of host/mock/*.h
(the reference's own headers need Sophus, Eigen, OpenCV, Boost and Pangolin, none of which exist in this image).  This test
is the compile-level evidence that can be had here: for every member of KeyFrame / MapPoint / Map / Frame / GeometricCamera
the adapter touches, the declaration is parsed out of the REFERENCE's header and out of the mock, and name, arity, parameter
types, constness and return / field type must agree.  Nothing of the reference is copied: its headers are read where they
lie, and the test is skipped where /root/reference does not exist (the GPU box).

A small declaration parser, not a C++ front end: comments and preprocessor lines go, the class body is cut into statements
at depth 0, a statement with a '(' is a method (return type, name, parameter types without names and default values,
trailing const), anything else a list of field declarators behind a common base type."""
import os
import re

import pytest

from conftest import ROOT

REF = "/root/reference/include"
MOCK = os.path.join(ROOT, "mov-slam_amd", "host", "mock")
ADAPTER = os.path.join(ROOT, "mov-slam_amd", "host", "Optimizer.cc")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers exist in the build container only")

# class -> (reference header, mock header)
HEADERS = {"KeyFrame": ("KeyFrame.h", "KeyFrame.h"), "MapPoint": ("MapPoint.h", "MapPoint.h"), "Map": ("Map.h", "Map.h"),
           "Frame": ("Frame.h", "Frame.h"), "GeometricCamera": ("CameraModels/GeometricCamera.h", "GeometricCamera.h")}

# every member of the reference's classes the adapter (or the mocks' own restatements of MapPoint.cc, which the adapter's
# results are compared with) relies on: SURVEY.md 8(b) "Layering", plus what the normal / depth write-back reads
MEMBERS = {
    "KeyFrame": ["SetPose", "GetPose", "GetCameraCenter", "GetRightCameraCenter", "GetVectorCovisibleKeyFrames", "EraseMapPointMatch",
                 "GetMapPointMatches", "isBad", "GetMap", "mnId", "mnBALocalForKF", "mnBAFixedForKF", "mnBAGlobalForKF", "mTcwGBA",
                 "mvKeys", "mvKeysUn", "mvKeysRight", "mvuRight", "mnScaleLevels", "mvScaleFactors", "mvInvLevelSigma2", "mpCamera", "mpCamera2",
                 "NLeft", "NRight", "mbf"],
    "MapPoint": ["SetWorldPos", "GetWorldPos", "GetNormal", "SetNormalVector", "GetReferenceKeyFrame", "GetObservations", "Observations",
                 "AddObservation", "EraseObservation", "GetIndexInKeyFrame", "SetBadFlag", "isBad", "UpdateNormalAndDepth", "GetMap",
                 "mnId", "mnBALocalForKF", "mnBAGlobalForKF", "mPosGBA"],
    "Map": ["EraseMapPoint", "GetAllKeyFrames", "GetAllMapPoints", "GetInitKFid", "GetOriginKF", "IncreaseChangeIndex", "mMutexMapUpdate",
            "msOptKFs", "msFixedKFs"],
    "Frame": ["SetPose", "GetPose", "N", "mvKeys", "mvpMapPoints", "mvbOutlier", "mpCamera"],
    "GeometricCamera": ["getParameter"],
}

# members of the mocks that exist for the tests' own use (filling the map, counting calls) or that INTEGRATION.md proposes as
# additions to the reference: not part of the reference's API, so nothing to conform to
MOCK_ONLY = {"nObservationCopies", "nObservationVisits", "ForEachObservation", "SetMinMaxDistance", "mGlobalMutex", "nCenterReads", "nPoseSets",
             "mTcw", "mvCovisible", "mbBad", "mpMap", "mWorldPos", "mObservations", "nErased", "nNormalUpdates", "nObs", "vErasedBy", "mpRefKF",
             "mNormalVector", "mfMinDistance", "mfMaxDistance", "mvKFs", "mvMPs", "mnInitKFid", "mpOriginKF", "mnChangeIdx", "mspErased",
             "mMutexMap", "mMutexPose", "mMutexConnections", "mMutexFeatures", "mMutexPos", "mvParameters", "mvpMapPoints@KeyFrame"}


def _strip(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = "\n".join(l for l in src.split("\n") if not l.lstrip().startswith("#"))
    return src


def _class_body(src, name):
    """text between the braces of `class name {...}` (the definition, not a forward declaration)"""
    for m in re.finditer(r"\bclass\s+" + name + r"\b[^;{]*\{", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        return src[m.end():i - 1]
    raise AssertionError(f"class {name} not found")


def _statements(body):
    """statements of a class body at brace depth 0; a function body ends its statement"""
    out, cur, depth, paren = [], "", 0, 0
    for ch in body:
        if ch == "{":
            depth += 1
            if depth == 1 and paren == 0 and "(" in cur:
                continue                    # inline function body: dropped, the declaration in front of it is kept
        if ch == "}":
            depth -= 1
            if depth == 0:
                if "(" in cur: out.append(cur); cur = ""
                else: cur += "}"            # brace initialiser of a field
                continue
        if depth > 0:
            if "(" not in cur: cur += ch    # inside a field's brace initialiser
            continue
        paren += {"(": 1, ")": -1}.get(ch, 0)
        if ch == ";" and paren == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    return [re.sub(r"\s+", " ", s).strip() for s in out if s.strip()]


def _split_top(s, sep=","):
    parts, cur, depth = [], "", 0
    for ch in s:
        depth += {"<": 1, "(": 1, "{": 1, ">": -1, ")": -1, "}": -1}.get(ch, 0)
        if ch == sep and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    return parts + [cur]


def _norm_type(t):
    t = re.sub(r"\b(inline|virtual|static|explicit|mutable)\b", " ", t)
    t = t.replace("Sophus::SE3<float>", "Sophus::SE3f").replace("std::", "")
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*([*&<>,])\s*", r"\1", t)
    # `long unsigned int` and friends: one spelling
    words = t.split(" ")
    if set(words) <= {"long", "unsigned", "int", "const"} and "long" in words:
        t = ("const " if "const" in words else "") + ("unsigned " if "unsigned" in words else "") + "long"
    return t


def _param_type(p):
    p = _split_top(p, "=")[0].strip()                  # default value
    m = re.match(r"^(.*?[\s*&])([A-Za-z_]\w*)$", p)    # trailing parameter name
    if m and m.group(1).strip() and m.group(2) not in ("int", "float", "double", "bool", "char", "long", "unsigned"):
        p = m.group(1)
    return _norm_type(p)


def declarations(header_path, cls):
    """name -> list of ('method', return type, [parameter types], const) / ('field', type)"""
    body = _class_body(_strip(open(header_path).read()), cls)
    out = {}
    for st in _statements(body):
        st = re.sub(r"^(public|protected|private)\s*:\s*", "", st).strip()
        st = re.sub(r"^(public|protected|private)\s*:\s*", "", st).strip()
        if not st or st.startswith(("friend", "typedef", "using", "template", "enum", "struct", "class", "EIGEN_")):
            continue
        top = "".join(ch if d == 0 else "" for ch, d in _depths(st))
        if "(" in st and (("=" not in top) or top.index("(") < top.index("=")) and not re.match(r"^[\w:<>\s,*&]+\s+\w+\s*=", top):
            m = re.match(r"^(.*?)([~A-Za-z_]\w*)\s*\((.*)\)\s*(const)?\s*(?:=\s*0|override|noexcept)?\s*$", st)
            if not m:
                continue
            ret, name, params, cst = m.group(1), m.group(2), m.group(3), bool(m.group(4))
            if name == cls or name.startswith("~") or not ret.strip():
                continue                                 # constructors / destructor
            plist = [] if params.strip() in ("", "void") else [_param_type(p) for p in _split_top(params)]
            out.setdefault(name, []).append(("method", _norm_type(ret), plist, cst))
        else:
            decls = _split_top(st)
            first = _split_top(decls[0], "=")[0].strip()
            m = re.match(r"^(.*?)([*&\s]+)([A-Za-z_]\w*)$", first)
            if not m:
                continue
            base = m.group(1).strip()
            names = [(m.group(2).strip(), m.group(3))]
            for d in decls[1:]:
                d = _split_top(d, "=")[0].strip()
                m2 = re.match(r"^([*&\s]*)([A-Za-z_]\w*)$", d)
                if m2: names.append((m2.group(1).strip(), m2.group(2)))
            for ptr, name in names:
                out.setdefault(name, []).append(("field", _norm_type(base + ptr)))
    return out


def _depths(s):
    d = 0
    for ch in s:
        if ch in ")>}": d -= 1
        yield ch, d
        if ch in "(<{": d += 1


@pytest.mark.parametrize("cls", sorted(HEADERS))
def test_mock_declarations_equal_the_references(cls):
    ref = declarations(os.path.join(REF, HEADERS[cls][0]), cls)
    mock = declarations(os.path.join(MOCK, HEADERS[cls][1]), cls)
    for name in MEMBERS[cls]:
        assert name in ref, f"{cls}::{name}: not declared by the reference's {HEADERS[cls][0]}"
        assert name in mock, f"{cls}::{name}: the adapter's list names it, the mock does not declare it"
        r, m = sorted(map(repr, ref[name])), sorted(map(repr, mock[name]))
        # (an overload the adapter does not call may be missing from the mock; what the mock declares must exist in the reference)
        for decl in m:
            assert decl in r, f"{cls}::{name}: mock declares {decl}, the reference {r}"


def test_every_member_the_adapter_touches_is_in_the_list():
    """A new accessor in the adapter (or a mock member it starts using) must enter MEMBERS, i.e. be checked against the reference."""
    src = _strip(open(ADAPTER).read())
    used = set(re.findall(r"(?:->|\.)\s*([A-Za-z_]\w*)", src))
    for cls, (_, mh) in HEADERS.items():
        mock = declarations(os.path.join(MOCK, mh), cls)
        for name in mock:
            if name in used and name not in MEMBERS[cls] and name not in MOCK_ONLY and f"{name}@{cls}" not in MOCK_ONLY:
                raise AssertionError(f"the adapter uses {cls}::{name}, which is neither checked against the reference nor marked mock-only")


def test_the_check_fails_when_a_mock_signature_is_edited(tmp_path):
    """The comparison has teeth: a mock whose GetObservations() returns another container, whose mvuRight changes its element type
    or whose EraseMapPointMatch loses its const reference no longer conforms."""
    ref = declarations(os.path.join(REF, "MapPoint.h"), "MapPoint")
    src = open(os.path.join(MOCK, "MapPoint.h")).read()
    bad = tmp_path / "MapPoint.h"
    bad.write_text(src.replace("std::map<KeyFrame *, std::tuple<int, int>> GetObservations()", "std::map<KeyFrame *, size_t> GetObservations()"))
    mock = declarations(str(bad), "MapPoint")
    assert repr(mock["GetObservations"][0]) not in map(repr, ref["GetObservations"])
    refk = declarations(os.path.join(REF, "KeyFrame.h"), "KeyFrame")
    srck = open(os.path.join(MOCK, "KeyFrame.h")).read()
    for old, new, name in (("std::vector<float> mvuRight", "std::vector<double> mvuRight", "mvuRight"),
                           ("void EraseMapPointMatch(const int &idx)", "void EraseMapPointMatch(int idx)", "EraseMapPointMatch"),
                           ("long unsigned int mnId = 0, mnBALocalForKF", "int mnId = 0, mnBALocalForKF", "mnBALocalForKF"),
                           ("Sophus::SE3f GetPose()", "Sophus::SE3f GetPose() const", "GetPose")):
        assert old in srck, old
        badk = tmp_path / "KeyFrame.h"
        badk.write_text(srck.replace(old, new))
        mk = declarations(str(badk), "KeyFrame")
        assert any(repr(d) not in map(repr, refk[name]) for d in mk[name]), name


def test_optimizer_header_keeps_the_references_five_signatures():
    """include/Optimizer.h:45-60: the five static methods, their default arguments and the alignment macro, statement by statement
    (Tracking.cc and LocalMapping.cc must compile unchanged against the drop-in)."""
    ref = _statements(_class_body(_strip(open(os.path.join(REF, "Optimizer.h")).read()), "Optimizer"))
    own = _statements(_class_body(_strip(open(os.path.join(ROOT, "mov-slam_amd", "host", "Optimizer.h")).read()), "Optimizer"))
    norm = lambda st: re.sub(r"\s*([*&<>,()=])\s*", r"\1", re.sub(r"^(public|protected|private)\s*:\s*", "", st)).strip()
    ref_n, own_n = [norm(s) for s in ref], [norm(s) for s in own]
    methods = [s for s in ref_n if "(" in s]
    assert len(methods) == 5, methods
    for s in ref_n:
        assert s in own_n, f"the drop-in's Optimizer.h lacks: {s}"
