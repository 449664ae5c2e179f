! dbcsr_resident_loop -- a Fortran host that keeps its matrices ON THE DEVICE across multiplies (this repository's own program;
! it only CALLS the reference's public routines and the dbcsr_amd_dev_* interface of dbcsr_amd/fortran/dbcsr_amd_resident.F).
!
! What linear-scaling SCF does with DBCSR -- purification and sign iterations multiply the same few matrices dozens of times --
! in its smallest form: nrep times C <- beta C + alpha A B with the performance driver's matrices (seed, block sizes, C then A
! then B from dbcsr_make_random_matrix, tests/dbcsr_performance_multiply.F:271-450), once through the library's own
! dbcsr_multiply (whatever path the build takes) and once with A, B, C uploaded ONCE, multiplied nrep times in HBM and C
! downloaded ONCE.  Printed: seconds per multiply of both, GFLOP/s of the resident loop (alone, and with the one upload and
! download), and the largest relative difference of the two results' checksums (same blocks, same values up to summation order).
!
!   dbcsr_resident_loop M sparsity block_size nrep [check [mode]]     (check = 0: skip the reference loop, time the resident one only)
! mode 0: C accumulates (C_in of a multiply is the previous product).
! mode 1: every multiply starts from the ORIGINAL C (the reference's own performance driver resets C before each repetition,
!         tests/dbcsr_performance_multiply.F:597-629): the product goes to a second resident matrix (dbcsr_amd_dev_multiply, c_out), A, B and
!         C_in are the same generation of the same device arrays in every call -- the engine's plan is reused by address (index stamps).
! mode 2: a purification-style chain X_1 = beta C + alpha A B, X_{n+1} = beta C + alpha X_n B: the PRODUCT of one multiply is the left
!         operand of the next, made one on the device side (dbcsr_amd_dev_as_operand: on several ranks its row panel is gathered from
!         the peers' tiles) -- no download, no dbcsr_type in between.
PROGRAM dbcsr_resident_loop
   USE dbcsr_amd_resident, ONLY: dbcsr_amd_dev_as_operand, dbcsr_amd_dev_create, dbcsr_amd_dev_download, dbcsr_amd_dev_multiply, &
                                 dbcsr_amd_dev_release, dbcsr_amd_dev_sync, dbcsr_amd_dev_type
   USE dbcsr_dist_methods, ONLY: dbcsr_distribution_new, dbcsr_distribution_release
   USE dbcsr_dist_operations, ONLY: dbcsr_dist_bin
   USE dbcsr_dist_util, ONLY: dbcsr_checksum
   USE dbcsr_kinds, ONLY: int_8, real_8
   USE dbcsr_lib, ONLY: dbcsr_finalize_lib, dbcsr_init_lib
   USE dbcsr_machine, ONLY: m_walltime
   USE dbcsr_methods, ONLY: dbcsr_get_num_blocks, dbcsr_release
   USE dbcsr_mp_methods, ONLY: dbcsr_mp_new, dbcsr_mp_release
   USE dbcsr_mpiwrap, ONLY: mp_cart_create, mp_cart_rank, mp_comm_free, mp_comm_type, mp_environ, mp_max, mp_sum, mp_sync, &
                            mp_world_finalize, mp_world_init
   USE dbcsr_multiply_api, ONLY: dbcsr_multiply
   USE dbcsr_operations, ONLY: dbcsr_copy
   USE dbcsr_test_methods, ONLY: dbcsr_make_random_block_sizes, dbcsr_make_random_matrix, dbcsr_reset_randmat_seed
   USE dbcsr_types, ONLY: dbcsr_distribution_obj, dbcsr_mp_obj, dbcsr_type, dbcsr_type_real_8
   IMPLICIT NONE

   CHARACTER(len=100) :: arg
   INTEGER :: m, bs, nrep, check, mode, irep, numnodes, mynode, npdims(2), myploc(2), row, col, nblk_c
   REAL(real_8) :: sparsity, alpha, beta, t0, t1, t_ref, t_up, t_loop, t_down, cs_ref, cs_dev, csp_ref, csp_dev
   REAL(real_8), ALLOCATABLE :: t_rep(:)
   INTEGER(int_8) :: flop, flop_total
   INTEGER, DIMENSION(:, :), POINTER :: pgrid
   INTEGER, DIMENSION(:), POINTER, CONTIGUOUS :: sizes, rd, cd
   TYPE(mp_comm_type) :: mp_comm, group
   TYPE(dbcsr_mp_obj) :: mp_env
   TYPE(dbcsr_distribution_obj) :: dist
   TYPE(dbcsr_type) :: ma, mb, mc, mc_ref, mc_dev, mx
   TYPE(dbcsr_amd_dev_type) :: da, db, dc, dx, dy
   REAL(real_8) :: t_op
   LOGICAL :: ok

   m = 2316; sparsity = 0.8_real_8; bs = 23; nrep = 4; check = 1; mode = 0
   IF (command_argument_count() >= 1) THEN
      CALL get_command_argument(1, arg); READ (arg, *) m
   END IF
   IF (command_argument_count() >= 2) THEN
      CALL get_command_argument(2, arg); READ (arg, *) sparsity
   END IF
   IF (command_argument_count() >= 3) THEN
      CALL get_command_argument(3, arg); READ (arg, *) bs
   END IF
   IF (command_argument_count() >= 4) THEN
      CALL get_command_argument(4, arg); READ (arg, *) nrep
   END IF
   IF (command_argument_count() >= 5) THEN
      CALL get_command_argument(5, arg); READ (arg, *) check
   END IF
   IF (command_argument_count() >= 6) THEN
      CALL get_command_argument(6, arg); READ (arg, *) mode
   END IF
   ! a contraction keeps the iterates bounded: C <- beta C + alpha A B with entries in (0, 1) grows by about fill * m / 4 per step
   alpha = 1.0_real_8/MAX(1.0_real_8, (1.0_real_8 - sparsity)*REAL(m, real_8)/4.0_real_8)
   beta = 0.5_real_8

   CALL mp_world_init(mp_comm)
   CALL mp_environ(numnodes, mynode, mp_comm)
   npdims(:) = 0
   CALL mp_cart_create(mp_comm, 2, npdims, myploc, group)
   ALLOCATE (pgrid(0:npdims(1) - 1, 0:npdims(2) - 1))
   DO row = 0, npdims(1) - 1
      DO col = 0, npdims(2) - 1
         CALL mp_cart_rank(group, (/row, col/), pgrid(row, col))
      END DO
   END DO
   CALL dbcsr_mp_new(mp_env, group, pgrid, mynode, numnodes, myprow=myploc(1), mypcol=myploc(2))
   DEALLOCATE (pgrid)
   CALL dbcsr_init_lib(mp_comm%get_handle(), 0)

   CALL dbcsr_reset_randmat_seed()
   CALL dbcsr_make_random_block_sizes(sizes, m, (/1, bs/))
   CALL make(mc, "Matrix C")
   CALL make(ma, "Matrix A")
   CALL make(mb, "Matrix B")

   t_ref = 0.0_real_8
   cs_ref = 0.0_real_8; csp_ref = 0.0_real_8
   IF (check /= 0) THEN
      CALL dbcsr_copy(mc_ref, mc)
      t0 = m_walltime()
      DO irep = 1, nrep
         IF (mode == 1) THEN   ! every repetition from the original C
            CALL dbcsr_copy(mc_ref, mc)
            CALL dbcsr_multiply('N', 'N', alpha, ma, mb, beta, mc_ref)
         ELSE IF (mode == 2 .AND. irep > 1) THEN   ! X_{n+1} = beta C + alpha X_n B
            CALL dbcsr_copy(mx, mc_ref)
            CALL dbcsr_copy(mc_ref, mc)
            CALL dbcsr_multiply('N', 'N', alpha, mx, mb, beta, mc_ref)
            CALL dbcsr_release(mx)
         ELSE
            CALL dbcsr_multiply('N', 'N', alpha, ma, mb, beta, mc_ref)
         END IF
      END DO
      t_ref = (m_walltime() - t0)/REAL(nrep, real_8)
      cs_ref = dbcsr_checksum(mc_ref); csp_ref = dbcsr_checksum(mc_ref, pos=.TRUE.)
   END IF

   ! upload once ...
   t0 = m_walltime()
   ! (several ranks: A also brings the blocks of its process row to this rank's device, B those of its process column -- once)
   CALL dbcsr_amd_dev_create(ma, da, ok, role='A')
   IF (ok) CALL dbcsr_amd_dev_create(mb, db, ok, role='B')
   IF (ok) CALL dbcsr_amd_dev_create(mc, dc, ok, role='C')
   IF (ok .AND. mode /= 0) CALL dbcsr_amd_dev_create(mc, dx, ok, role='C')   ! (receives products: its blocks are dropped)
   IF (ok .AND. mode == 2) CALL dbcsr_amd_dev_create(mc, dy, ok, role='C')
   IF (.NOT. ok) STOP "dbcsr_resident_loop: upload failed"
   CALL mp_sync(group)
   t_up = m_walltime() - t0
   ! ... multiply in HBM ...
   flop_total = 0
   t_op = 0.0_real_8
   ALLOCATE (t_rep(nrep))
   t0 = m_walltime()
   DO irep = 1, nrep
      t1 = m_walltime()
      IF (mode == 1) THEN
         CALL dbcsr_amd_dev_multiply('N', 'N', alpha, da, db, beta, dc, ok, flop=flop, c_out=dx)
      ELSE IF (mode == 2 .AND. irep == 1) THEN
         CALL dbcsr_amd_dev_multiply('N', 'N', alpha, da, db, beta, dc, ok, flop=flop, c_out=dx)
      ELSE IF (mode == 2) THEN
         ! the previous product becomes the left operand (several ranks: its row panel is gathered), the new one goes to the other matrix
         IF (MOD(irep, 2) == 0) THEN
            CALL dbcsr_amd_dev_as_operand(dx, 'A', ok)
            t_op = t_op + m_walltime() - t1
            IF (ok) CALL dbcsr_amd_dev_multiply('N', 'N', alpha, dx, db, beta, dc, ok, flop=flop, c_out=dy)
         ELSE
            CALL dbcsr_amd_dev_as_operand(dy, 'A', ok)
            t_op = t_op + m_walltime() - t1
            IF (ok) CALL dbcsr_amd_dev_multiply('N', 'N', alpha, dy, db, beta, dc, ok, flop=flop, c_out=dx)
         END IF
      ELSE
         CALL dbcsr_amd_dev_multiply('N', 'N', alpha, da, db, beta, dc, ok, flop=flop)
      END IF
      IF (.NOT. ok) STOP "dbcsr_resident_loop: device multiply failed"
      CALL dbcsr_amd_dev_sync()
      CALL mp_sync(group)   ! (a multiply is over when the slowest rank's is)
      t_rep(irep) = m_walltime() - t1
      CALL mp_sum(flop, group)
      flop_total = flop_total + flop
   END DO
   t_loop = m_walltime() - t0
   ! ... download once
   t0 = m_walltime()
   CALL dbcsr_copy(mc_dev, mc)
   IF (mode == 0) THEN
      CALL dbcsr_amd_dev_download(dc, mc_dev, ok)
   ELSE IF (mode == 1 .OR. MOD(nrep, 2) == 1) THEN
      CALL dbcsr_amd_dev_download(dx, mc_dev, ok)
   ELSE
      CALL dbcsr_amd_dev_download(dy, mc_dev, ok)
   END IF
   IF (.NOT. ok) STOP "dbcsr_resident_loop: download failed"
   CALL mp_sync(group)
   t_down = m_walltime() - t0
   cs_dev = dbcsr_checksum(mc_dev); csp_dev = dbcsr_checksum(mc_dev, pos=.TRUE.)
   nblk_c = dbcsr_get_num_blocks(mc_dev)
   CALL mp_sum(nblk_c, group)

   IF (mynode == 0) THEN
   WRITE (*, '(A,I0,A,F6.3,A,I0,A,I0,A,I0,A,I0,A,I0)') " resident_loop: M ", m, "  sparsity ", sparsity, "  block ", bs, "  multiplies ", nrep, &
      "  ranks ", numnodes, " = ", npdims(1), " x ", npdims(2)
   WRITE (*, '(A,I0,A,I0)') " resident_loop: blocks of C ", nblk_c, "  flop ", flop_total
   IF (mode == 1) THEN
      WRITE (*, '(A)') " resident_loop: mode 1 (C_in = the ORIGINAL C in every multiply, product out of place)"
   ELSE IF (mode == 2) THEN
      WRITE (*, '(A)') " resident_loop: mode 2 (chain: the product is the next multiply's left operand, dbcsr_amd_dev_as_operand)"
   ELSE
      WRITE (*, '(A)') " resident_loop: mode 0 (C accumulates: C_in = the previous product)"
   END IF
   IF (mode == 2) WRITE (*, '(A,F10.5)') " resident_loop: product -> operand, per multiply [s] ", t_op/REAL(MAX(1, nrep - 1), real_8)
   WRITE (*, '(A,F10.5,A,F10.5,A,F10.5)') " resident_loop: upload once [s] ", t_up, "  download once [s] ", t_down, &
      "  per multiply [s] ", t_loop/REAL(nrep, real_8)
   WRITE (*, '(A,F12.3)') " resident_loop: GFLOP/s of the multiplies in HBM ", REAL(flop_total, real_8)/t_loop*1.0E-9_real_8
   ! the first multiplies build their plan and make the allocator grow (C's pattern changes once: sparse C_in -> product pattern);
   ! from the third on the plan and the buffers are reused
   WRITE (*, '(A,20(1X,F8.5))') " resident_loop: seconds of each multiply ", t_rep(1:MIN(nrep, 20))
   IF (nrep >= 3) WRITE (*, '(A,F10.5,A,F12.3)') " resident_loop: steady state (third multiply on) per multiply [s] ", &
      SUM(t_rep(3:nrep))/REAL(nrep - 2, real_8), "  GFLOP/s ", REAL(flop, real_8)/(SUM(t_rep(3:nrep))/REAL(nrep - 2, real_8))*1.0E-9_real_8
   WRITE (*, '(A,F12.3)') " resident_loop: GFLOP/s with the one upload and the one download ", &
      REAL(flop_total, real_8)/(t_up + t_loop + t_down)*1.0E-9_real_8
   IF (check /= 0) THEN
      WRITE (*, '(A,F10.5)') " resident_loop: dbcsr_multiply of this build, per multiply [s] ", t_ref
      WRITE (*, '(A,2(1X,ES23.15E3))') " resident_loop: checksums reference ", cs_ref, csp_ref
      WRITE (*, '(A,2(1X,ES23.15E3))') " resident_loop: checksums resident  ", cs_dev, csp_dev
      WRITE (*, '(A,ES10.3)') " resident_loop: relative difference ", MAX(ABS(cs_dev - cs_ref)/MAX(ABS(cs_ref), 1.0E-300_real_8), &
                                                                    ABS(csp_dev - csp_ref)/MAX(ABS(csp_ref), 1.0E-300_real_8))
   END IF
   END IF

   CALL dbcsr_amd_dev_release(da); CALL dbcsr_amd_dev_release(db); CALL dbcsr_amd_dev_release(dc)
   IF (mode /= 0) CALL dbcsr_amd_dev_release(dx)
   IF (mode == 2) CALL dbcsr_amd_dev_release(dy)
   CALL dbcsr_release(ma); CALL dbcsr_release(mb); CALL dbcsr_release(mc); CALL dbcsr_release(mc_dev)
   IF (check /= 0) CALL dbcsr_release(mc_ref)
   DEALLOCATE (sizes)
   CALL dbcsr_mp_release(mp_env)
   CALL mp_comm_free(group)
   CALL dbcsr_finalize_lib()
   CALL mp_world_finalize()

CONTAINS

   SUBROUTINE make(mat, name)
      TYPE(dbcsr_type), INTENT(OUT) :: mat
      CHARACTER(len=*), INTENT(IN) :: name
      CALL dbcsr_dist_bin(rd, SIZE(sizes), npdims(1), sizes)
      CALL dbcsr_dist_bin(cd, SIZE(sizes), npdims(2), sizes)
      CALL dbcsr_distribution_new(dist, mp_env, rd, cd)
      CALL dbcsr_make_random_matrix(mat, sizes, sizes, name, sparsity, group, data_type=dbcsr_type_real_8, dist=dist)
      CALL dbcsr_distribution_release(dist)
      DEALLOCATE (rd, cd)
   END SUBROUTINE make

END PROGRAM dbcsr_resident_loop
