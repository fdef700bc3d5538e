"""One rank of the CF_GPUS=N stage script on the host-emulated kernels (tests/test_sharded_cli.py): the package's own rank entry
(centroflye_amd/sharded_cli.py: rank_main) bound to tests/emu/libcfhip_emu.so, whose file transport wants a directory as the
rendezvous.  Started by sharded_cli.launch(..., rank_cmd=[python, this file]) with RANK / WORLD_SIZE / CF_COMM_ID_FILE set."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from centroflye_amd import _lib, sharded_cli  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load(os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so"))
    # the emulator's file transport wants a DIRECTORY as the rendezvous: the launcher's file name (sharded_cli.launch) or, under a
    # launcher that only exports RANK / WORLD_SIZE (torch.distributed.run), the name the ranks derive from their common parent
    from centroflye_amd.sharded import default_rendezvous
    rdv = default_rendezvous()
    os.environ["CF_COMM_ID_FILE"] = rdv
    os.makedirs(rdv, exist_ok=True)
    sys.exit(sharded_cli.rank_main(sys.argv[1:], lib=lib, device=0, sub_edges=int(os.environ.get("CF_TEST_SUB_EDGES", "0")) or None,
                                   knobs={"dist_slots": 2048, "dist_block": 128}))
