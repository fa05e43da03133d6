"""Headless render driver:  python -m termdaw_amd <project_dir> [--scan] [-o out.wav]
                        python -m termdaw_amd <project_dir> --stream [--realtime] [-o out.wav] < events

The reference renders only from its TUI (`render` / `normalize` commands, ui_workflow.rs:120-133); the first form is
the same sequence -- State::refresh, optionally State::scan_exact, State::render -- without the TUI.
<project_dir> holds project.toml ([settings] main, buffer_length, project_samplerate) and the project script.

--stream is the reference's stream workflow (stream_workflow.rs:41-105) without the audio device: events for the
streams the script declared (declare_stream) arrive on stdin, blocks are pulled one at a time at the playhead.
The floww crate's binary packets are not restated; the wire format here is text, one event per line,
    <stream name> <t_sec> <note> <vel>
a blank line ends a packet (trim_streams, append, set_time(graph time) -- stream_workflow.rs:62-69), after which the
blocks up to that packet's latest event time are pulled; `end <t_sec>` (or EOF) renders on to <t_sec> / one more block
and stops.  --realtime paces the pulls like the reference (half a second ahead of the wall clock); without it the
stream is rendered as fast as the events arrive.  The pulled blocks are written as a 16-bit WAV.
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
    ap.add_argument("--exact-bandpass", action="store_true",
                    help="band-pass vertices through the exact kernels (bit-identical to the reference's recurrence; default: scan mode, "
                         "<= 1e-6 RMS / +-1 LSB, 28x faster on a deep effect chain)")
    ap.add_argument("--exact-sine", action="store_true",
                    help="debug_sine / synth vertices evaluate glibc's sinf bit for bit (default: the tolerance-class device sine, <= 1e-6 RMS; "
                         "config 3's oscillators take 0.32 instead of 0.09 ms)")
    ap.add_argument("--stream", action="store_true", help="stream workflow: events from stdin, block pulls at the playhead")
    ap.add_argument("--realtime", action="store_true", help="with --stream: pace the pulls against the wall clock")
    args = ap.parse_args(argv)
    api.set_device(args.device)
    s = api.State(open_dir=args.project_dir)
    if args.exact_bandpass:
        s.set_option("band_mode", 0)
    if args.exact_sine:
        s.set_option("sine_mode", 1)
    if args.stream:
        return stream(s, args)
    t0 = time.perf_counter()
    if not s.refresh():
        print("TermDaw: refresh failed: %s" % api.last_error(), file=sys.stderr)
        return 1
    t1 = time.perf_counter()
    if args.scan:
        s.scan_exact()
    s.render(args.output)
    t2 = time.perf_counter()
    print("Ok: rendered %d blocks to %s (%d-bit, %d Hz): load %.1f ms, render+write %.1f ms"
          % (s.cs, args.output or s.output_file, s.bd, s.render_sr, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    return 0


def stream(s, args, lines=None):
    """stream_workflow.rs:41-105 on text events (see the module docstring).  `lines` defaults to sys.stdin."""
    import numpy as np
    from .workloads import write_wav_int16
    if not s.refresh():
        print("TermDaw: refresh failed: %s" % api.last_error(), file=sys.stderr)
        return 1
    g, sb, fb = s.g, s.sb, s.fb
    bl, sr = g.bl, g.sr
    blocks = []
    t_start = time.perf_counter()

    def pull_until(t_sec):
        while g.get_time() < int(t_sec * sr):
            if args.realtime:   # half a second ahead of the wall clock (stream_workflow.rs:88-90)
                ahead = g.get_time() / sr - (time.perf_counter() - t_start)
                if ahead > 0.5:
                    time.sleep(ahead - 0.5)
            fb.set_time(g.get_time())                      # stream_workflow.rs:91-92
            out = g.render(sb, fb)
            if out is None:
                raise api.TermdawError(api.last_error() or "no output vertex")
            blocks.append(np.stack(out, axis=1))
            fb.set_time_to_next_block()

    packet, latest, end_at = {}, 0.0, None
    def flush():
        if not packet:
            return
        fb.trim_streams()                                  # stream_workflow.rs:64-68
        for name, ev in packet.items():
            if fb.append_stream(name, ev) < 0:
                print("MSGs: unknown stream %r" % name, file=sys.stderr)
        fb.set_time(g.get_time())
        packet.clear()

    for line in (lines if lines is not None else sys.stdin):
        tok = line.split()
        if not tok:
            flush()
            pull_until(latest)
            continue
        if tok[0] == "end":
            end_at = float(tok[1]) if len(tok) > 1 else latest
            break
        name, t, note, vel = tok[0], float(tok[1]), float(tok[2]), float(tok[3])
        packet.setdefault(name, []).append((t, note, vel))
        latest = max(latest, t)
    flush()
    pull_until(max(latest, end_at or 0.0) + bl / sr)
    f = np.concatenate(blocks) if blocks else np.zeros((0, 2), np.float32)
    # the WAV sink's quantiser (state.rs:517-521): (x * 32767) as i16 -- truncating, saturating, NaN -> 0
    q = np.nan_to_num(f.astype(np.float32) * np.float32(32767.0), nan=0.0, posinf=32767.0, neginf=-32768.0)
    pcm = np.clip(np.trunc(q), -32768, 32767).astype(np.int16)
    out = args.output or s.output_file
    # the pulled blocks are at the PROJECT rate (the stream workflow plays them as they are, stream_workflow.rs:92-101):
    # the header carries that rate, never set_render_samplerate()'s
    write_wav_int16(out, pcm, sr)
    print("Ok: streamed %d blocks (%.2f s) to %s in %.1f ms" % (len(blocks), len(blocks) * bl / sr, out,
                                                               (time.perf_counter() - t_start) * 1e3))
    return 0


if __name__ == "__main__":
    sys.exit(main())
