"""rsparse_amd -- host-side mirror of rsparse's numeric solve path over librsparse_hip.so (MI355X / gfx950).

Same function names, argument meaning and error behaviour as the reference's public API for this path
(/root/reference/src/lib.rs): lsolve, ltsolve, usolve, utsolve, schol, chol, cholsol, sqr, lu, lusol, qr, qrsol,
gaxpy on `Sprs` matrices (and `Sprs.save` / `Sprs.load`, src/data.rs:414-516).  Symbolic analysis runs on the host inside the library, numeric work on the GPU.
There is no CPU fallback: importing fails loudly if the HIP library has not been built
(`make -C rsparse_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`).
"""
from .api import (  # noqa: F401
    CholHandle,
    DeviceError,
    Error,
    LuHandle,
    MatHandle,
    NoPivot,
    Nmrc,
    NotPositiveDefinite,
    Sprs,
    Symb,
    TriHandle,
    program_checksum,
    validate_program,
    validate_refill,
    cache_stats,
    cache_clear,
    amd,
    chol,
    cholsol,
    cholsol_batch,
    device_count,
    gaxpy,
    lib,
    lsolve,
    ltsolve,
    lu,
    lu_static_pattern,
    lusol,
    qr,
    qrsol,
    transpose,
    QrHandle,
    schol,
    set_devices,
    solve_pair_dev,
    sqr,
    usolve,
    utsolve,
)
