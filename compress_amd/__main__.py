"""linzip (reference main.go) over the MI355X path: python -m compress_amd -i FILE [-d] [-o OUT] [-no_out] [-r]
[-dict DICT] [-version].  Same flags, messages and exit codes as the reference CLI."""
import argparse
import sys

EXTENSION = ".linzip"   # main.go:23-26
VERSION = "0.3.0"


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="linzip", add_help=True, prefix_chars="-")
    ap.add_argument("-d", action="store_true", help="decompress")
    ap.add_argument("-i", default="", help="input file (required)")
    ap.add_argument("-o", default="", help="output file")
    ap.add_argument("-no_out", action="store_true", help="no output")
    ap.add_argument("-r", action="store_true", help="report compression ratio")
    ap.add_argument("-dict", default="", help="compression dictionary")
    ap.add_argument("-version", action="store_true", help="report executable version")
    a = ap.parse_args(argv)

    def quit_(msg: str) -> int:   # main.go:28-33
        sys.stderr.write(msg)
        return 1

    if a.version:
        print("linzip v" + VERSION)
        return 0
    if a.i == "":
        return quit_("no input file specified\n")
    try:
        data = open(a.i, "rb").read()
        dict_ = open(a.dict, "rb").read() if a.dict else b""
    except OSError as e:
        return quit_(f"{e}\n")
    if a.o != "" and a.no_out:
        return quit_("options -no_out and -o are mutually exclusive\n")
    out_name = a.o
    if out_name == "":            # main.go:66-76
        if a.d:
            out_name = a.i[: -len(EXTENSION)] if a.i.endswith(EXTENSION) else a.i + ".decompressed"
        else:
            out_name = a.i + EXTENSION
    import compress_amd
    try:
        if a.d:
            out = compress_amd.Decompress(data, dict_)
            len_c, len_d = len(data), len(out)
        else:
            out = compress_amd.NewCompressor(dict_).Compress(data)
            len_c, len_d = len(out), len(data)
    except Exception as e:        # assertNoError, main.go:35-39
        return quit_(f"{e}\n")
    if not a.no_out:
        try:
            with open(out_name, "wb") as f:
                f.write(out)
        except OSError as e:
            return quit_(f"{e}\n")
    if a.r:                       # main.go:98-101
        pct = len_d * 100 // len_c
        print("%d B -> %d B compression ratio %d.%02d" % (len(data), len(out), pct // 100, pct % 100))
    return 0


if __name__ == "__main__":
    sys.exit(main())
