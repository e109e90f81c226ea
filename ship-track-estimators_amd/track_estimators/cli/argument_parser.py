"""
Command-line surface of ``track_estimator``.

The flag names, destinations and defaults are the reference's (cli/argument_parser.py:14-91 there), because scripts such
as its examples/cli_example/run.sh must keep working; they are declared here as one table.  ``--no-noise`` and a
comma-separated ``-s`` are additive extras of this build.
"""
import argparse

from .. import __version__

# (short, long, dest, kwargs)
_OPTIONS = (
    ("-i", "--input", "input_file", dict(default="input.json", help="Filepath to the input JSON file")),
    ("-o", "--output", "output_prefix", dict(default="output", help="Output file prefix")),
    ("-t", "--track-file", "track_file", dict(required=True, help="Filepath to the ship track data")),
    ("-s", "--ship-id", "ship_id", dict(required=True, help="Ship ID; several comma-separated IDs are filtered in one "
                                                           "batched GPU launch")),
    ("-lat", "--latitude-id", "lat_id", dict(required=True, help="Name of the latitude column")),
    ("-lon", "--longitude-id", "lon_id", dict(required=True, help="Name of the longitude column")),
    ("-ic", "--id-col", "id_col", dict(required=True, help="Name of the ship ID column")),
    ("-rts", "--rts-smoother", "apply_rts_smoother", dict(action="store_true",
                                                          help="Apply the Rauch-Tung-Striebel (RTS) smoother")),
    ("-rev", "--reverse", "reverse", dict(action="store_true", help="Reverse the trajectory")),
    (None, "--no-noise", "no_noise", dict(action="store_true", help="Skip the process/measurement noise the reference "
                                                                    "injects (deterministic output)")),
)


def create_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     description=f"Ship track estimator {__version__} command line interface")
    for short, long_, dest, kw in _OPTIONS:
        names = [n for n in (short, long_) if n]
        parser.add_argument(*names, dest=dest, **kw)
    parser.add_argument("-v", "--version", action="version", version=f"%(prog)s {__version__}")
    return parser
