"""INTEGRATION.md section 1 (the maintainer's CMake change) checked against the reference package's own build file:
the facade's obstacle.h / cf_manager.h must shadow the reference's headers for `panda_bimanual_control_node` ONLY --
src/obstacle.cpp and src/cf_agent.cpp are also compiled into dual_panda_costp_controller, vrep_interface,
dynamic_obstacle_node and vision_interface_node (B/CMakeLists.txt:105-116, :165-175, :191-195, :210-214), which must
keep the reference's headers. CPU tests; the ones that read /root/reference skip themselves where it is absent (the
GPU box)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = "/root/reference/src/bimanual_planning_ros"
NODE = "panda_bimanual_control_node"
REPLACED_SOURCES = {"src/obstacle.cpp", "src/cf_agent.cpp", "src/cf_manager.cpp", "src/helper_functions.cpp"}
FACADE_SHADOWS = {"cf_manager.h", "obstacle.h"}   # the only header names the facade may share with B/include

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(B, "CMakeLists.txt")),
                                     reason="reference checkout not present (GPU box)")


def cmake_commands(text):
    """[(command, [arguments])] of a CMake file / snippet (comments stripped; no nested parentheses in this package)"""
    text = re.sub(r"#[^\n]*", "", text)
    return [(m.group(1).lower(), m.group(2).split()) for m in re.finditer(r"(\w+)\s*\(([^()]*)\)", text)]


def targets_with_sources(cmds):
    return {a[0]: [x for x in a[1:] if x.startswith("src/")] for c, a in cmds if c in ("add_library", "add_executable") and a}


def recipe_snippet():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 1. What the maintainer changes"):md.index("## 2.")]
    blocks = re.findall(r"```cmake\n(.*?)```", sec, re.S)
    assert len(blocks) == 1
    return blocks[0]


def test_recipe_is_scoped_to_the_planner_node():
    cmds = cmake_commands(recipe_snippet())
    names = [c for c, _ in cmds]
    # nothing directory-wide
    for banned in ("include_directories", "add_definitions", "add_compile_definitions", "add_compile_options", "link_libraries"):
        assert banned not in names, "directory-wide command in the recipe: " + banned
    for c, a in cmds:
        if c.startswith("target_") or c in ("add_executable", "add_dependencies"):
            assert a[0] == NODE, (c, a)
    inc = [a for c, a in cmds if c == "target_include_directories"]
    assert inc == [[NODE, "BEFORE", "PRIVATE", "${PMAF_ROOT}/include"]]
    assert [a for c, a in cmds if c == "target_compile_definitions"] == [[NODE, "PRIVATE", "PMAF_USE_EIGEN"]]
    link = [a for c, a in cmds if c == "target_link_libraries"]
    assert len(link) == 1 and "pmaf_hip" in link[0]
    # the imported library is the in-tree one
    props = [a for c, a in cmds if c == "set_target_properties"]
    assert props and props[0][0] == "pmaf_hip" and props[0][-1].endswith("predictive-multi-agent-framework_amd/lib/libpmaf_hip.so")


@needs_reference
def test_recipe_source_list_is_the_reference_list_minus_the_replaced_units():
    ref = targets_with_sources(cmake_commands(open(os.path.join(B, "CMakeLists.txt")).read()))
    mine = targets_with_sources(cmake_commands(recipe_snippet()))
    assert set(mine) == {NODE, "pmaf_hip"} and mine["pmaf_hip"] == []   # (the IMPORTED library has no sources)
    assert set(ref[NODE]) - set(mine[NODE]) == REPLACED_SOURCES
    assert set(mine[NODE]) <= set(ref[NODE])
    assert set(ref[NODE]) & REPLACED_SOURCES == REPLACED_SOURCES


@needs_reference
def test_other_targets_compiling_obstacle_or_agent_keep_the_reference_headers():
    """every target of the reference package that compiles obstacle.cpp / cf_agent.cpp / cf_manager.cpp, and which of
    them the recipe touches (the node, nothing else)"""
    ref = targets_with_sources(cmake_commands(open(os.path.join(B, "CMakeLists.txt")).read()))
    users = {t for t, srcs in ref.items() if set(srcs) & {"src/obstacle.cpp", "src/cf_agent.cpp", "src/cf_manager.cpp"}}
    assert users == {NODE, "dual_panda_costp_controller", "vrep_interface", "dynamic_obstacle_node", "vision_interface_node"}
    touched = {a[0] for c, a in cmake_commands(recipe_snippet()) if c.startswith("target_") or c == "add_executable"}
    assert touched == {NODE}
    # the directory-wide include path of the package is `include` alone (:84-86): the recipe adds to the node's own list
    dirwide = [a for c, a in cmake_commands(open(os.path.join(B, "CMakeLists.txt")).read()) if c == "include_directories"]
    assert dirwide == [["include"]]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for t in users - {NODE}:
        assert t in md, "INTEGRATION.md must say that %s keeps the reference's headers" % t


def _resolve(name, search):
    for d in search:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


@needs_reference
def test_node_include_set_resolves_to_the_facade_for_two_headers_only():
    """the node target's include order [PMAF/include, B/include]: walking the <bimanual_planning_ros/...> includes of
    the node's remaining sources, cf_manager.h (panda_bimanual_control.h:6) and obstacle.h (:9) come from the facade,
    everything else from the reference, and neither cf_agent.h nor helper_functions.h -- whose translation units the
    recipe drops -- is reached"""
    search = [os.path.join(ROOT, "include"), os.path.join(B, "include")]
    mine = targets_with_sources(cmake_commands(recipe_snippet()))[NODE]
    facade_names = set(os.listdir(os.path.join(ROOT, "include", "bimanual_planning_ros")))
    ref_names = set(os.listdir(os.path.join(B, "include", "bimanual_planning_ros")))
    assert facade_names & ref_names == FACADE_SHADOWS
    seen, from_facade, queue = set(), set(), [os.path.join(B, s) for s in mine]
    generated = set()   # message headers (catkin generates them: Position.h, Obstacles.h ...)
    while queue:
        f = queue.pop()
        if f in seen:
            continue
        seen.add(f)
        for m in re.finditer(r'#include\s*[<"](bimanual_planning_ros/[\w.]+)[>"]', open(f).read()):
            r = _resolve(m.group(1), search)
            if r is None:
                generated.add(m.group(1))
                continue
            if r.startswith(search[0]):
                from_facade.add(os.path.basename(r))
            queue.append(r)
    assert from_facade == FACADE_SHADOWS, from_facade
    reached = {os.path.basename(f) for f in seen}
    assert "panda_bimanual_control.h" in reached
    assert not ({"cf_agent.h", "helper_functions.h"} & reached), reached
    msgs = {os.path.splitext(f)[0] for f in os.listdir(os.path.join(B, "msg"))} | {os.path.splitext(f)[0] for f in os.listdir(os.path.join(B, "srv"))}
    assert {os.path.splitext(os.path.basename(g))[0] for g in generated} <= msgs, generated


@needs_reference
def test_compiler_picks_the_facade_with_the_node_include_order():
    """the same through a compiler: `g++ -M` of the two includes the node header makes, with the node target's include
    order and the API-shape Eigen declarations (tests/cpp/eigen_api_check; this image has no Eigen3), lists the facade's
    headers and none of the reference's"""
    chk = os.path.join(ROOT, "tests", "cpp", "eigen_api_check")
    src = "#include <bimanual_planning_ros/cf_manager.h>\n#include <bimanual_planning_ros/obstacle.h>\n"
    r = subprocess.run(["g++", "-std=c++17", "-M", "-DPMAF_USE_EIGEN", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(B, "include"), "-I" + chk, "-x", "c++", "-"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    deps = r.stdout.decode().replace("\\\n", " ").split()
    ours = [d for d in deps if "bimanual_planning_ros/" in d]
    assert ours and all(os.path.abspath(d).startswith(os.path.join(ROOT, "include")) for d in ours), ours
    # and the node's call forms still compile in that order (syntax only)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DPMAF_USE_EIGEN",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(B, "include"), "-I" + chk,
                        "-I" + os.path.join(chk, "eigen3"), os.path.join(chk, "node_style_caller.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


@needs_reference
def test_recipe_through_cmake_on_a_package_with_the_reference_targets(tmp_path):
    """the recipe run by CMake itself: a package with the reference's targets and source lists (names read from
    B/CMakeLists.txt; the sources are probes that only include <bimanual_planning_ros/obstacle.h>, the package's own
    `include/` holds a marker header in the reference's place), its directory-wide include_directories(include), and
    INTEGRATION.md's snippet verbatim in place of the node's add_executable. From CMake's compile_commands.json: the
    node's four translation units -- and no other, not even the SAME source file compiled into another target
    (parameter_manager.cpp: also in dynamic_obstacle_node and vision_interface_node) -- get -DPMAF_USE_EIGEN and this repository's include/ IN FRONT
    OF the package's; run with exactly those commands the node's probes see the facade's header and every other
    target's the package's own."""
    import json
    import shutil
    if shutil.which("cmake") is None:
        pytest.skip("no cmake")
    ref = targets_with_sources(cmake_commands(open(os.path.join(B, "CMakeLists.txt")).read()))
    pkg = tmp_path / "pkg"
    (pkg / "include" / "bimanual_planning_ros").mkdir(parents=True)
    (pkg / "src").mkdir()
    (pkg / "include" / "bimanual_planning_ros" / "obstacle.h").write_text("#pragma once\n#define PACKAGE_OWN_OBSTACLE_H 1\n")
    node_sources = set(targets_with_sources(cmake_commands(recipe_snippet()))[NODE])
    for srcs in ref.values():   # one probe for every source: which obstacle.h a translation unit sees is read back below
        for src in srcs:
            (pkg / src).write_text("#include <bimanual_planning_ros/obstacle.h>\n#ifdef PACKAGE_OWN_OBSTACLE_H\nint saw_package_header;\n"
                                   "#else\nstatic ghostplanner::cfplanner::Obstacle saw_facade_header;\n#endif\n")
    lines = ["cmake_minimum_required(VERSION 3.10)", "project(bimanual_planning_ros CXX)", "set(CMAKE_EXPORT_COMPILE_COMMANDS ON)",
             "set(CMAKE_CXX_STANDARD 17)", "include_directories(include)"]
    for t, srcs in ref.items():
        if t == NODE:
            continue
        kind = "add_library(%s STATIC" if t in ("utilities", "dual_panda_costp_controller") else "add_executable(%s"
        lines.append((kind % t) + " " + " ".join(srcs) + ")")
    snippet = recipe_snippet().replace("set(PMAF_ROOT /opt/pmaf)", "set(PMAF_ROOT %s)" % ROOT)
    assert ROOT in snippet
    lines.append(snippet)
    (pkg / "CMakeLists.txt").write_text("\n".join(lines) + "\n")
    bld = tmp_path / "build"
    r = subprocess.run(["cmake", "-S", str(pkg), "-B", str(bld), "-G", "Unix Makefiles"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    cc = json.load(open(bld / "compile_commands.json"))
    mine_inc, pkg_inc = "-I" + os.path.join(ROOT, "include"), "-I" + str(pkg / "include")
    chk = os.path.join(ROOT, "tests", "cpp", "eigen_api_check")
    seen_node = set()
    for e in cc:
        rel = os.path.relpath(e["file"], str(pkg))
        args = e["command"].split()
        is_node = "CMakeFiles/%s.dir/" % NODE in e["command"]
        assert ("-DPMAF_USE_EIGEN" in args) == is_node, (rel, e["command"])
        assert (mine_inc in args) == is_node, (rel, e["command"])
        assert pkg_inc in args
        if is_node:
            seen_node.add(rel)
            assert args.index(mine_inc) < args.index(pkg_inc), e["command"]
        # preprocess / check the probe with CMake's own command (+ the API-shape Eigen declarations: no Eigen3 here)
        out = args[args.index("-o") + 1]
        cmd = [a for a in args if a not in ("-o", "-c", out)] + ["-I" + chk]
        c = subprocess.run(cmd + ["-fsyntax-only"], capture_output=True, text=True, cwd=e["directory"])
        assert c.returncode == 0, (rel, c.stderr[-1500:])
        c = subprocess.run(cmd + ["-E", "-P"], capture_output=True, text=True, cwd=e["directory"])
        assert c.returncode == 0, (rel, c.stderr[-1500:])
        assert ("saw_facade_header" in c.stdout) == is_node and ("saw_package_header" in c.stdout) == (not is_node), rel
    assert seen_node == node_sources
