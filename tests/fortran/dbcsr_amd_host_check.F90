! Fortran host check: ISO_C_BINDING -> C-ABI -> HIP, compiled with amdflang.
!
! Uses the REFERENCE'S OWN binding module for devices (src/acc/dbcsr_acc_device.F, compiled from the
! reference checkout where it lies by oracle/build_ref.sh -- it needs no fypp) and, for the entry points whose
! reference modules cannot be built here (they pull in dbcsr_config -> dbcsr_mpiwrap -> fypp), interface blocks
! with the signatures of src/acc/acc.h:46-73 and src/acc/acc_libsmm.h:42-47 as the reference declares them in
! src/acc/dbcsr_acc_stream.F, dbcsr_acc_devmem.F and src/mm/dbcsr_acc_operations.F:38-53.
! It runs one parameter stack of 23x23x23 products through libsmm_acc_transpose + libsmm_acc_process and
! compares with MATMUL.  Exit code 0 = pass.
MODULE dbcsr_amd_c_abi
   USE ISO_C_BINDING, ONLY: C_INT, C_SIZE_T, C_PTR, C_CHAR, C_INT32_T, C_INT64_T, C_DOUBLE
   IMPLICIT NONE
   TYPE, BIND(C) :: dbcsr_amd_bcsr   ! struct dbcsr_amd_bcsr of include/dbcsr_amd_mm.h (device pointers)
      INTEGER(C_INT32_T) :: nblkrows, nblkcols
      TYPE(C_PTR)        :: row_blk_size, col_blk_size, row_p, col_i, blk_p, data
      INTEGER(C_INT64_T) :: nblks
      INTEGER(C_INT64_T) :: index_stamp = 0
   END TYPE
   INTERFACE
      FUNCTION acc_init() RESULT(istat) BIND(C, name="c_dbcsr_acc_init")
         IMPORT; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION acc_finalize() RESULT(istat) BIND(C, name="c_dbcsr_acc_finalize")
         IMPORT; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION smm_init() RESULT(istat) BIND(C, name="libsmm_acc_init")
         IMPORT; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION smm_is_thread_safe() RESULT(yes) BIND(C, name="libsmm_acc_is_thread_safe")
         IMPORT; INTEGER(C_INT) :: yes
      END FUNCTION
      FUNCTION stream_create(stream_ptr, name, priority) RESULT(istat) BIND(C, name="c_dbcsr_acc_stream_create")
         IMPORT; TYPE(C_PTR) :: stream_ptr; CHARACTER(KIND=C_CHAR), DIMENSION(*) :: name
         INTEGER(C_INT), VALUE :: priority; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION stream_destroy(stream_ptr) RESULT(istat) BIND(C, name="c_dbcsr_acc_stream_destroy")
         IMPORT; TYPE(C_PTR), VALUE :: stream_ptr; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION stream_sync(stream_ptr) RESULT(istat) BIND(C, name="c_dbcsr_acc_stream_sync")
         IMPORT; TYPE(C_PTR), VALUE :: stream_ptr; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION dev_mem_alloc(mem, n) RESULT(istat) BIND(C, name="c_dbcsr_acc_dev_mem_allocate")
         IMPORT; TYPE(C_PTR) :: mem; INTEGER(C_SIZE_T), VALUE :: n; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION dev_mem_dealloc(mem) RESULT(istat) BIND(C, name="c_dbcsr_acc_dev_mem_deallocate")
         IMPORT; TYPE(C_PTR), VALUE :: mem; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION memcpy_h2d(host, dev, n, stream_ptr) RESULT(istat) BIND(C, name="c_dbcsr_acc_memcpy_h2d")
         IMPORT; TYPE(C_PTR), VALUE :: host, dev, stream_ptr; INTEGER(C_SIZE_T), VALUE :: n; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION memcpy_d2h(dev, host, n, stream_ptr) RESULT(istat) BIND(C, name="c_dbcsr_acc_memcpy_d2h")
         IMPORT; TYPE(C_PTR), VALUE :: dev, host, stream_ptr; INTEGER(C_SIZE_T), VALUE :: n; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION smm_transpose(trs_stack, offset, nblks, buffer, data_type, m, n, max_kernel_dim, stream_ptr) &
         RESULT(istat) BIND(C, name="libsmm_acc_transpose")
         IMPORT; TYPE(C_PTR), VALUE :: trs_stack, buffer, stream_ptr
         INTEGER(C_INT), VALUE :: offset, nblks, data_type, m, n, max_kernel_dim; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION smm_process(param_stack_host, param_stack_dev, stack_size, data_type, a_data, b_data, c_data, m_max, n_max, &
                           k_max, max_kernel_dim, def_mnk, stack_stream_ptr, c_stream_ptr) RESULT(istat) BIND(C, name="libsmm_acc_process")
         IMPORT; TYPE(C_PTR), VALUE :: param_stack_host, param_stack_dev, a_data, b_data, c_data, stack_stream_ptr, c_stream_ptr
         INTEGER(C_INT), VALUE :: stack_size, data_type, m_max, n_max, k_max, max_kernel_dim, def_mnk; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION mm_create(handle) RESULT(istat) BIND(C, name="dbcsr_amd_mm_create")
         IMPORT; TYPE(C_PTR) :: handle; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION mm_destroy(handle) RESULT(istat) BIND(C, name="dbcsr_amd_mm_destroy")
         IMPORT; TYPE(C_PTR), VALUE :: handle; INTEGER(C_INT) :: istat
      END FUNCTION
      ! the whole operator of one rank in one call (include/dbcsr_amd_mm.h)
      FUNCTION amd_multiply(handle, transa, transb, datatype, alpha, a, b, beta, c, limits, retain_sparsity, filter_eps, &
                            c_out, flop, stream) RESULT(istat) BIND(C, name="dbcsr_amd_multiply")
         IMPORT; TYPE(C_PTR), VALUE :: handle, stream; CHARACTER(KIND=C_CHAR), VALUE :: transa, transb
         INTEGER(C_INT), VALUE :: datatype, retain_sparsity; REAL(C_DOUBLE), VALUE :: alpha, beta, filter_eps
         TYPE(dbcsr_amd_bcsr), INTENT(IN) :: a, b, c; INTEGER(C_INT64_T), DIMENSION(6), INTENT(IN) :: limits
         TYPE(dbcsr_amd_bcsr) :: c_out; INTEGER(C_INT64_T) :: flop; INTEGER(C_INT) :: istat
      END FUNCTION
      FUNCTION bcsr_release(m) RESULT(istat) BIND(C, name="dbcsr_amd_bcsr_release")
         IMPORT; TYPE(dbcsr_amd_bcsr) :: m; INTEGER(C_INT) :: istat
      END FUNCTION
   END INTERFACE
END MODULE dbcsr_amd_c_abi

PROGRAM dbcsr_amd_host_check
   USE ISO_C_BINDING, ONLY: C_INT, C_SIZE_T, C_PTR, C_LOC, C_NULL_PTR, C_NULL_CHAR, C_INT32_T, C_DOUBLE
   USE dbcsr_acc_device, ONLY: dbcsr_acc_get_ndevices, dbcsr_acc_set_active_device, dbcsr_acc_clear_errors  ! reference module
   USE dbcsr_amd_c_abi
   IMPLICIT NONE
   INTEGER, PARAMETER :: m = 23, n = 23, k = 23, na = 40, nb = 50, nc = 7, nstack = 600, dbcsr_type_real_8 = 3
   REAL(C_DOUBLE), ALLOCATABLE, TARGET :: a(:), b(:), c(:), cref(:)
   INTEGER(C_INT32_T), ALLOCATABLE, TARGET :: stack(:, :), trs(:)
   TYPE(C_PTR) :: stream, da, db, dc, ds, dt
   INTEGER :: s, ia, ib, ic, ndev, istat, i
   REAL(C_DOUBLE) :: err, r, err2
   INTERFACE
      SUBROUTINE check_native_multiply(err_out)
         USE ISO_C_BINDING, ONLY: C_DOUBLE
         REAL(C_DOUBLE), INTENT(OUT) :: err_out
      END SUBROUTINE
   END INTERFACE
   REAL(C_DOUBLE) :: ablk(m, k), bblk(k, n), cblk(m, n)

   ndev = dbcsr_acc_get_ndevices()          ! reference: src/acc/dbcsr_acc_device.F
   IF (ndev < 1) STOP 2
   CALL dbcsr_acc_set_active_device(0)
   IF (acc_init() /= 0) STOP 3
   IF (smm_init() /= 0) STOP 4
   IF (smm_is_thread_safe() /= 1) STOP 5
   ALLOCATE (a(na*m*k), b(nb*k*n), c(nc*m*n), cref(nc*m*n), stack(3, nstack), trs(nb))
   CALL RANDOM_NUMBER(a); CALL RANDOM_NUMBER(b); CALL RANDOM_NUMBER(c)
   cref = c
   DO s = 1, nstack                          ! stack sorted by c offset, 1-based element offsets (dbcsr_mm_types.F:24-37)
      CALL RANDOM_NUMBER(r); ia = INT(r*na)
      CALL RANDOM_NUMBER(r); ib = INT(r*nb)
      ic = ((s - 1)*nc)/nstack
      stack(:, s) = (/ia*m*k + 1, ib*k*n + 1, ic*m*n + 1/)
      ablk = RESHAPE(a(ia*m*k + 1:(ia + 1)*m*k), (/m, k/))
      bblk = RESHAPE(b(ib*k*n + 1:(ib + 1)*k*n), (/k, n/))
      cblk = MATMUL(ablk, bblk)
      cref(ic*m*n + 1:(ic + 1)*m*n) = cref(ic*m*n + 1:(ic + 1)*m*n) + RESHAPE(cblk, (/m*n/))
   END DO
   DO i = 1, nb
      trs(i) = (i - 1)*k*n                   ! 0-based offsets (dbcsr_mm_common.F:412)
   END DO
   stream = C_NULL_PTR
   IF (stream_create(stream, "host_check"//C_NULL_CHAR, -1) /= 0) STOP 6
   IF (dev_mem_alloc(da, INT(8*SIZE(a), C_SIZE_T)) /= 0) STOP 7
   IF (dev_mem_alloc(db, INT(8*SIZE(b), C_SIZE_T)) /= 0) STOP 7
   IF (dev_mem_alloc(dc, INT(8*SIZE(c), C_SIZE_T)) /= 0) STOP 7
   IF (dev_mem_alloc(ds, INT(4*SIZE(stack), C_SIZE_T)) /= 0) STOP 7
   IF (dev_mem_alloc(dt, INT(4*SIZE(trs), C_SIZE_T)) /= 0) STOP 7
   IF (memcpy_h2d(C_LOC(a), da, INT(8*SIZE(a), C_SIZE_T), stream) /= 0) STOP 8
   IF (memcpy_h2d(C_LOC(b), db, INT(8*SIZE(b), C_SIZE_T), stream) /= 0) STOP 8
   IF (memcpy_h2d(C_LOC(c), dc, INT(8*SIZE(c), C_SIZE_T), stream) /= 0) STOP 8
   IF (memcpy_h2d(C_LOC(stack), ds, INT(4*SIZE(stack), C_SIZE_T), stream) /= 0) STOP 8
   IF (memcpy_h2d(C_LOC(trs), dt, INT(4*SIZE(trs), C_SIZE_T), stream) /= 0) STOP 8
   ! what multiply_cannon does per tick: transpose the B panel (dbcsr_mm_cannon.F:1625), then process the stacks
   IF (smm_transpose(dt, 0, nb, db, dbcsr_type_real_8, k, n, 80, stream) /= 0) STOP 9
   istat = smm_process(C_LOC(stack), ds, nstack, dbcsr_type_real_8, da, db, dc, m, n, k, 80, 1, stream, stream)
   IF (istat < 0) STOP 10
   IF (memcpy_d2h(dc, C_LOC(c), INT(8*SIZE(c), C_SIZE_T), stream) /= 0) STOP 11
   IF (stream_sync(stream) /= 0) STOP 12
   err = MAXVAL(ABS(c - cref)/MAX(ABS(cref), 1.0D-300))
   istat = dev_mem_dealloc(da) + dev_mem_dealloc(db) + dev_mem_dealloc(dc) + dev_mem_dealloc(ds) + dev_mem_dealloc(dt)
   istat = istat + stream_destroy(stream)
   CALL check_native_multiply(err2)
   WRITE (*, '(A,ES10.3)') "fortran host check: dbcsr_amd_multiply max abs err vs MATMUL = ", err2
   IF (err2 > 1.0D-12) STOP 15
   CALL dbcsr_acc_clear_errors()
   IF (acc_finalize() /= 0) STOP 13
   WRITE (*, '(A,I0,A,ES10.3)') "fortran host check: devices=", ndev, " max rel err vs MATMUL = ", err
   IF (err > 1.0D-10 .OR. istat /= 0) STOP 14
END PROGRAM dbcsr_amd_host_check

! Second part: dbcsr_amd_multiply through its ISO_C_BINDING interface on small block-sparse matrices built in Fortran
! (1-based Fortran arrays, 0-based BCSR index as the C-ABI wants it), checked against MATMUL on dense copies.
SUBROUTINE check_native_multiply(err_out)
   USE ISO_C_BINDING
   USE dbcsr_amd_c_abi
   IMPLICIT NONE
   REAL(C_DOUBLE), INTENT(OUT) :: err_out
   INTEGER, PARAMETER :: nb = 5, nfull = 21
   INTEGER(C_INT32_T), TARGET :: sizes(nb) = (/3, 5, 2, 7, 4/)
   INTEGER :: off(nb + 1), i
   TYPE host_mat
      INTEGER(C_INT32_T), ALLOCATABLE :: row_p(:), col_i(:)
      INTEGER(C_INT64_T), ALLOCATABLE :: blk_p(:)
      REAL(C_DOUBLE), ALLOCATABLE :: dat(:)
      REAL(C_DOUBLE) :: dense(nfull, nfull)
   END TYPE
   TYPE(host_mat), TARGET :: ha, hb, hc, hr
   TYPE(dbcsr_amd_bcsr) :: da, db, dc, dr
   TYPE(C_PTR) :: dsizes, handle
   INTEGER(C_INT64_T) :: limits(6), flop, nze
   REAL(C_DOUBLE), PARAMETER :: alpha = 0.5D0, beta = -2.0D0
   REAL(C_DOUBLE) :: expect(nfull, nfull)
   INTEGER :: r, c, b, j

   off(1) = 0
   DO i = 1, nb
      off(i + 1) = off(i) + sizes(i)
   END DO
   CALL build(ha, 1); CALL build(hb, 2); CALL build(hc, 3)
   IF (dev_mem_alloc(dsizes, INT(4*nb, C_SIZE_T)) /= 0) STOP 20
   IF (memcpy_h2d(C_LOC(sizes), dsizes, INT(4*nb, C_SIZE_T), C_NULL_PTR) /= 0) STOP 20
   CALL upload(ha, da); CALL upload(hb, db); CALL upload(hc, dc)
   handle = C_NULL_PTR
   IF (mm_create(handle) /= 0) STOP 21
   limits = 0
   IF (amd_multiply(handle, 'N', 'N', 3, alpha, da, db, beta, dc, limits, 0, 0.0D0, dr, flop, C_NULL_PTR) /= 0) STOP 22
   ! download the result
   ALLOCATE (hr%row_p(nb + 1), hr%col_i(dr%nblks), hr%blk_p(dr%nblks))
   IF (memcpy_d2h(dr%row_p, C_LOC(hr%row_p), INT(4*(nb + 1), C_SIZE_T), C_NULL_PTR) /= 0) STOP 23
   IF (memcpy_d2h(dr%col_i, C_LOC(hr%col_i), INT(4*dr%nblks, C_SIZE_T), C_NULL_PTR) /= 0) STOP 23
   IF (memcpy_d2h(dr%blk_p, C_LOC(hr%blk_p), INT(8*dr%nblks, C_SIZE_T), C_NULL_PTR) /= 0) STOP 23
   IF (stream_sync(C_NULL_PTR) /= 0) STOP 23
   nze = 0
   DO r = 1, nb
      DO b = hr%row_p(r) + 1, hr%row_p(r + 1)
         nze = nze + sizes(r)*sizes(hr%col_i(b) + 1)
      END DO
   END DO
   ALLOCATE (hr%dat(nze))
   IF (memcpy_d2h(dr%data, C_LOC(hr%dat), INT(8*nze, C_SIZE_T), C_NULL_PTR) /= 0) STOP 23
   IF (stream_sync(C_NULL_PTR) /= 0) STOP 23
   hr%dense = 0.0D0
   DO r = 1, nb
      DO b = hr%row_p(r) + 1, hr%row_p(r + 1)
         c = hr%col_i(b) + 1
         DO j = 1, sizes(c)
            DO i = 1, sizes(r)
               hr%dense(off(r) + i, off(c) + j) = hr%dat(hr%blk_p(b) + i + sizes(r)*(j - 1))
            END DO
         END DO
      END DO
   END DO
   expect = beta*hc%dense + alpha*MATMUL(ha%dense, hb%dense)
   err_out = MAXVAL(ABS(hr%dense - expect))
   IF (bcsr_release(dr) /= 0) STOP 24
   IF (mm_destroy(handle) /= 0) STOP 24
CONTAINS
   SUBROUTINE build(h, which)
      TYPE(host_mat), INTENT(INOUT) :: h
      INTEGER, INTENT(IN) :: which
      INTEGER :: rr, cc, nblk, ii, jj
      INTEGER(C_INT64_T) :: pos
      LOGICAL :: keep
      ALLOCATE (h%row_p(nb + 1), h%col_i(nb*nb), h%blk_p(nb*nb), h%dat(nfull*nfull))
      h%dense = 0.0D0
      nblk = 0; pos = 0
      DO rr = 1, nb
         h%row_p(rr) = nblk
         DO cc = 1, nb
            SELECT CASE (which)
            CASE (1); keep = MOD(rr + cc, 2) == 0
            CASE (2); keep = MOD(rr*cc + 1, 3) /= 0
            CASE DEFAULT; keep = rr == cc
            END SELECT
            IF (.NOT. keep) CYCLE
            nblk = nblk + 1
            h%col_i(nblk) = cc - 1
            h%blk_p(nblk) = pos
            DO jj = 1, sizes(cc)
               DO ii = 1, sizes(rr)
                  pos = pos + 1
                  h%dat(pos) = REAL(MOD(7*ii + 3*jj + 11*rr + 5*cc + which, 17), C_DOUBLE)/8.0D0 - 1.0D0
                  h%dense(off(rr) + ii, off(cc) + jj) = h%dat(pos)
               END DO
            END DO
         END DO
      END DO
      h%row_p(nb + 1) = nblk
   END SUBROUTINE
   SUBROUTINE upload(h, d)
      TYPE(host_mat), INTENT(IN), TARGET :: h
      TYPE(dbcsr_amd_bcsr), INTENT(OUT) :: d
      INTEGER :: nblk
      nblk = h%row_p(nb + 1)
      d%nblkrows = nb; d%nblkcols = nb; d%nblks = nblk
      d%row_blk_size = dsizes; d%col_blk_size = dsizes
      IF (dev_mem_alloc(d%row_p, INT(4*(nb + 1), C_SIZE_T)) /= 0) STOP 25
      IF (dev_mem_alloc(d%col_i, INT(4*nb*nb, C_SIZE_T)) /= 0) STOP 25
      IF (dev_mem_alloc(d%blk_p, INT(8*nb*nb, C_SIZE_T)) /= 0) STOP 25
      IF (dev_mem_alloc(d%data, INT(8*nfull*nfull, C_SIZE_T)) /= 0) STOP 25
      IF (memcpy_h2d(C_LOC(h%row_p), d%row_p, INT(4*(nb + 1), C_SIZE_T), C_NULL_PTR) /= 0) STOP 26
      IF (memcpy_h2d(C_LOC(h%col_i), d%col_i, INT(4*nb*nb, C_SIZE_T), C_NULL_PTR) /= 0) STOP 26
      IF (memcpy_h2d(C_LOC(h%blk_p), d%blk_p, INT(8*nb*nb, C_SIZE_T), C_NULL_PTR) /= 0) STOP 26
      IF (memcpy_h2d(C_LOC(h%dat), d%data, INT(8*nfull*nfull, C_SIZE_T), C_NULL_PTR) /= 0) STOP 26
   END SUBROUTINE
END SUBROUTINE check_native_multiply
