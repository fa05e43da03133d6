"""Headless render driver:  python -m termdaw_amd <project_dir> [--scan] [-o out.wav]

The reference renders only from its TUI (`render` / `normalize` commands, ui_workflow.rs:120-133); this is the
same sequence -- State::refresh, optionally State::scan_exact, State::render -- without the TUI.
<project_dir> holds project.toml ([settings] main, buffer_length, project_samplerate) and the project script.
"""
import argparse
import sys
import time

from . import api


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m termdaw_amd", description=__doc__.split("\n")[0])
    ap.add_argument("project_dir")
    ap.add_argument("--scan", action="store_true", help="run the exact normalisation scan before rendering")
    ap.add_argument("-o", "--output", default=None, help="override set_output_file()")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    api.set_device(args.device)
    s = api.State(open_dir=args.project_dir)
    t0 = time.perf_counter()
    if not s.refresh():
        print("TermDaw: refresh failed: %s" % api.last_error(), file=sys.stderr)
        return 1
    t1 = time.perf_counter()
    if args.scan:
        s.scan_exact()
    s.render(args.output)
    t2 = time.perf_counter()
    frames = s.cs * 1024
    print("Ok: rendered %d blocks to %s (%d-bit, %d Hz): load %.1f ms, render+write %.1f ms"
          % (s.cs, args.output or s.output_file, s.bd, s.render_sr, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    return 0


if __name__ == "__main__":
    sys.exit(main())
