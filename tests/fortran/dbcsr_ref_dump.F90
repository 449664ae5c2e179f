! dbcsr_ref_dump -- fixture generator: runs the UNCHANGED reference library (built by tools/build_dbcsr_host.py) on one
! multiply case and writes everything a parity test needs: the block index of C (sorted), its checksums, the flop count
! and, for small cases, every value of C.  This is this repository's own program; it only CALLS the reference's public
! routines, in the order its performance driver does (tests/dbcsr_performance_multiply.F:271-450: seed reset, block
! sizes, C then A then B from dbcsr_make_random_matrix), so that the inputs are the ones the CPU oracle regenerates
! from (seed, counter).  Unlike the performance driver it passes the matrix symmetries to the generator, filter_eps
! to dbcsr_multiply, and 0 limits mean "not given".
!
!   dbcsr_ref_dump case.nml out.txt        (namelist &spec, see tools/make_ref_fixtures.py)
PROGRAM dbcsr_ref_dump
   USE dbcsr_dist_methods, ONLY: dbcsr_distribution_new, dbcsr_distribution_release
   USE dbcsr_dist_operations, ONLY: dbcsr_dist_bin
   USE dbcsr_dist_util, ONLY: dbcsr_checksum
   USE dbcsr_iterator_operations, ONLY: dbcsr_iterator_blocks_left, dbcsr_iterator_next_block, &
                                        dbcsr_iterator_start, dbcsr_iterator_stop
   USE dbcsr_kinds, ONLY: int_8, real_4, real_8
   USE dbcsr_lib, ONLY: dbcsr_finalize_lib, dbcsr_init_lib
   USE dbcsr_methods, ONLY: dbcsr_get_num_blocks, dbcsr_nblkcols_total, dbcsr_nblkrows_total, dbcsr_release
   USE dbcsr_mp_methods, ONLY: dbcsr_mp_new, dbcsr_mp_release
   USE dbcsr_mpiwrap, ONLY: mp_cart_create, mp_cart_rank, mp_comm_free, mp_comm_type, mp_environ, &
                            mp_world_finalize, mp_world_init
   USE dbcsr_multiply_api, ONLY: dbcsr_multiply
   USE dbcsr_test_methods, ONLY: dbcsr_make_random_block_sizes, dbcsr_make_random_matrix, dbcsr_reset_randmat_seed
   USE dbcsr_types, ONLY: dbcsr_distribution_obj, dbcsr_iterator, dbcsr_mp_obj, dbcsr_type, dbcsr_type_real_4, dbcsr_type_real_8
   IMPLICIT NONE

   INTEGER, PARAMETER :: maxbs = 16
   INTEGER :: m, n, k, limits(6), nbs_m, nbs_n, nbs_k, bs_m(2*maxbs), bs_n(2*maxbs), bs_k(2*maxbs), dump_values, data_type
   REAL(real_8) :: sp_a, sp_b, sp_c, alpha, beta, filter_eps
   CHARACTER :: transa, transb, symm_a, symm_b, symm_c
   LOGICAL :: retain
   NAMELIST /spec/ m, n, k, sp_a, sp_b, sp_c, transa, transb, symm_a, symm_b, symm_c, alpha, beta, limits, retain, &
      filter_eps, nbs_m, nbs_n, nbs_k, bs_m, bs_n, bs_k, dump_values, data_type

   CHARACTER(len=1000) :: fin, fout
   INTEGER :: numnodes, mynode, npdims(2), myploc(2), u, row, col, nblk
   INTEGER, DIMENSION(:, :), POINTER :: pgrid
   INTEGER, DIMENSION(:), POINTER, CONTIGUOUS :: sizes_m, sizes_n, sizes_k, rd, cd, rd2, cd2
   TYPE(mp_comm_type) :: mp_comm, group
   TYPE(dbcsr_mp_obj) :: mp_env
   TYPE(dbcsr_distribution_obj) :: dist
   TYPE(dbcsr_type) :: ma, mb, mc
   TYPE(dbcsr_iterator) :: iter
   REAL(real_8), DIMENSION(:, :), POINTER :: blk
   REAL(real_4), DIMENSION(:, :), POINTER :: blk4
   REAL(real_4) :: alpha4, beta4
   LOGICAL :: tr
   INTEGER(int_8) :: flop
   REAL(real_8) :: cs, cs_pos

   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   transa = 'N'; transb = 'N'; symm_a = 'N'; symm_b = 'N'; symm_c = 'N'
   alpha = 1.0_real_8; beta = 1.0_real_8; limits = 0; retain = .FALSE.; filter_eps = -1.0_real_8
   bs_m = 0; bs_n = 0; bs_k = 0; dump_values = 0; data_type = 3   ! 1 = real(4), 3 = real(8), as in the .perf files
   OPEN (newunit=u, file=TRIM(fin), status='old', action='read')
   READ (u, nml=spec)
   CLOSE (u)

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
   CALL dbcsr_make_random_block_sizes(sizes_m, m, bs_m(1:2*nbs_m))
   CALL dbcsr_make_random_block_sizes(sizes_n, n, bs_n(1:2*nbs_n))
   CALL dbcsr_make_random_block_sizes(sizes_k, k, bs_k(1:2*nbs_k))

   ! C (m x n), then A (op(A) is m x k), then B (op(B) is k x n): the generator's counter advances in this order
   CALL make(mc, sizes_m, sizes_n, "Matrix C", sp_c, symm_c)
   IF (transa .NE. 'N') THEN
      CALL make(ma, sizes_k, sizes_m, "Matrix A", sp_a, symm_a)
   ELSE
      CALL make(ma, sizes_m, sizes_k, "Matrix A", sp_a, symm_a)
   END IF
   IF (transb .NE. 'N') THEN
      CALL make(mb, sizes_n, sizes_k, "Matrix B", sp_b, symm_b)
   ELSE
      CALL make(mb, sizes_k, sizes_n, "Matrix B", sp_b, symm_b)
   END IF

   flop = 0
   alpha4 = REAL(alpha, real_4); beta4 = REAL(beta, real_4)
   IF (data_type == 1) THEN
      CALL multiply_real4()
   ELSE IF (ANY(limits .NE. 0)) THEN
      IF (filter_eps .GE. 0.0_real_8) THEN
         CALL dbcsr_multiply(transa, transb, alpha, ma, mb, beta, mc, first_row=lim(1), last_row=lim(2), first_column=lim(3), &
                             last_column=lim(4), first_k=lim(5), last_k=lim(6), retain_sparsity=retain, filter_eps=filter_eps, flop=flop)
      ELSE
         CALL dbcsr_multiply(transa, transb, alpha, ma, mb, beta, mc, first_row=lim(1), last_row=lim(2), first_column=lim(3), &
                             last_column=lim(4), first_k=lim(5), last_k=lim(6), retain_sparsity=retain, flop=flop)
      END IF
   ELSE
      IF (filter_eps .GE. 0.0_real_8) THEN
         CALL dbcsr_multiply(transa, transb, alpha, ma, mb, beta, mc, retain_sparsity=retain, filter_eps=filter_eps, flop=flop)
      ELSE
         CALL dbcsr_multiply(transa, transb, alpha, ma, mb, beta, mc, retain_sparsity=retain, flop=flop)
      END IF
   END IF
   cs = dbcsr_checksum(mc)
   cs_pos = dbcsr_checksum(mc, pos=.TRUE.)

   nblk = dbcsr_get_num_blocks(mc)
   IF (numnodes > 1) WRITE (fout, '(A,A,I0)') TRIM(fout), '.rank', mynode   ! (one file per rank: its own blocks, the global checksums)
   OPEN (newunit=u, file=TRIM(fout), status='replace', action='write')
   WRITE (u, '(A,3(1X,I0))') 'dims', dbcsr_nblkrows_total(mc), dbcsr_nblkcols_total(mc), nblk
   WRITE (u, '(A,1X,I0)') 'flop', flop
   WRITE (u, '(A,2(1X,ES24.16E3))') 'checksum', cs, cs_pos
   WRITE (u, '(A,2(1X,ES24.16E3))') 'checksum_a', dbcsr_checksum(ma), dbcsr_checksum(ma, pos=.TRUE.)
   WRITE (u, '(A,2(1X,ES24.16E3))') 'checksum_b', dbcsr_checksum(mb), dbcsr_checksum(mb, pos=.TRUE.)
   CALL dbcsr_iterator_start(iter, mc)
   DO WHILE (dbcsr_iterator_blocks_left(iter))
      IF (data_type == 1) THEN   ! single precision values are written as doubles (exactly representable)
         CALL dbcsr_iterator_next_block(iter, row, col, blk4, tr)
         WRITE (u, '(A,2(1X,I0),1X,L1,2(1X,I0))') 'block', row, col, tr, SIZE(blk4, 1), SIZE(blk4, 2)
         IF (dump_values .NE. 0) WRITE (u, '(4(1X,ES24.16E3))') REAL(blk4, real_8)
         CYCLE
      END IF
      CALL dbcsr_iterator_next_block(iter, row, col, blk, tr)
      IF (dump_values .NE. 0) THEN
         ! a block stored transposed (symmetric storage) is written as stored, with its flag
         WRITE (u, '(A,2(1X,I0),1X,L1,2(1X,I0))') 'block', row, col, tr, SIZE(blk, 1), SIZE(blk, 2)
         WRITE (u, '(4(1X,ES24.16E3))') blk
      ELSE
         WRITE (u, '(A,2(1X,I0),1X,L1,2(1X,I0))') 'block', row, col, tr, SIZE(blk, 1), SIZE(blk, 2)
      END IF
   END DO
   CALL dbcsr_iterator_stop(iter)
   CLOSE (u)

   CALL dbcsr_release(ma)
   CALL dbcsr_release(mb)
   CALL dbcsr_release(mc)
   DEALLOCATE (sizes_m, sizes_n, sizes_k)
   CALL dbcsr_mp_release(mp_env)
   CALL mp_comm_free(group)
   CALL dbcsr_finalize_lib()
   CALL mp_world_finalize()

CONTAINS

   INTEGER FUNCTION lim(i)
      INTEGER, INTENT(IN) :: i
      lim = limits(i)
   END FUNCTION lim

   SUBROUTINE multiply_real4()
      IF (ANY(limits .NE. 0)) THEN
         IF (filter_eps .GE. 0.0_real_8) THEN
            CALL dbcsr_multiply(transa, transb, alpha4, ma, mb, beta4, mc, first_row=lim(1), last_row=lim(2), first_column=lim(3), &
                                last_column=lim(4), first_k=lim(5), last_k=lim(6), retain_sparsity=retain, filter_eps=filter_eps, flop=flop)
         ELSE
            CALL dbcsr_multiply(transa, transb, alpha4, ma, mb, beta4, mc, first_row=lim(1), last_row=lim(2), first_column=lim(3), &
                                last_column=lim(4), first_k=lim(5), last_k=lim(6), retain_sparsity=retain, flop=flop)
         END IF
      ELSE
         IF (filter_eps .GE. 0.0_real_8) THEN
            CALL dbcsr_multiply(transa, transb, alpha4, ma, mb, beta4, mc, retain_sparsity=retain, filter_eps=filter_eps, flop=flop)
         ELSE
            CALL dbcsr_multiply(transa, transb, alpha4, ma, mb, beta4, mc, retain_sparsity=retain, flop=flop)
         END IF
      END IF
   END SUBROUTINE multiply_real4

   SUBROUTINE make(mat, rs, cs_, name, sparsity, symm)
      TYPE(dbcsr_type), INTENT(OUT) :: mat
      INTEGER, DIMENSION(:), POINTER, CONTIGUOUS :: rs, cs_
      CHARACTER(len=*), INTENT(IN) :: name
      REAL(real_8), INTENT(IN) :: sparsity
      CHARACTER, INTENT(IN) :: symm
      CALL dbcsr_dist_bin(rd, SIZE(rs), npdims(1), rs)
      CALL dbcsr_dist_bin(cd, SIZE(cs_), npdims(2), cs_)
      CALL dbcsr_distribution_new(dist, mp_env, rd, cd)
      CALL dbcsr_make_random_matrix(mat, rs, cs_, name, sparsity, group, &
                                    data_type=MERGE(dbcsr_type_real_4, dbcsr_type_real_8, data_type == 1), symmetry=symm, dist=dist)
      CALL dbcsr_distribution_release(dist)
      DEALLOCATE (rd, cd)
   END SUBROUTINE make

END PROGRAM dbcsr_ref_dump
