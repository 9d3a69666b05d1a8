"""CPU: `bench.py --gpus N` started WITHOUT a launcher spawns its own ranks (face_mask_inpaint_amd/launch.py): every child gets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / one shared free port, rank 0's stdout is relayed, a failing rank makes the parent fail,
and under torchrun (WORLD_SIZE already set) nothing is spawned."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, n=3):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent(body))
    drv = tmp_path / "parent.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from face_mask_inpaint_amd import launch
        assert launch.needs_spawn({n})
        sys.exit(launch.spawn_ranks({str(script)!r}, ["--tag", "x"], {n}, timeout_s=60))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(drv)], env=env, capture_output=True, text=True, timeout=120)


def test_spawn_ranks_relays_rank0_and_sets_the_rendezvous(tmp_path):
    r = _run(tmp_path, """
        import os, sys, json
        assert sys.argv[1:] == ["--tag", "x"]
        e = os.environ
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["LOCAL_RANK"] == e["RANK"] and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print(json.dumps({"rank": int(e["RANK"]), "world": int(e["WORLD_SIZE"]), "port": int(e["MASTER_PORT"])}))
    """)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and '"rank": 0' in lines[0] and '"world": 3' in lines[0]  # only rank 0 speaks on stdout


def test_spawn_ranks_fails_when_a_rank_fails(tmp_path):
    r = _run(tmp_path, """
        import os, sys
        sys.stderr.write("boom from %s\\n" % os.environ["RANK"])
        sys.exit(3 if os.environ["RANK"] == "2" else 0)
    """)
    assert r.returncode != 0 and "[rank 2] boom from 2" in r.stderr and "ranks failed" in r.stderr


def test_no_spawn_under_a_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    from face_mask_inpaint_amd import launch

    monkeypatch.setenv("WORLD_SIZE", "8")
    assert not launch.needs_spawn(8)
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("RANK", raising=False)
    assert launch.needs_spawn(2) and not launch.needs_spawn(1)
