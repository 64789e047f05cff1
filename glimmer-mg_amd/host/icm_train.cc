//  icm_train.cc -- ICM_Training_t of icm.hh: build-icm's training with the counting on the device.
//
//  The reference (src/ICM/icm.cc:1356-1455 Train_Model, 1061-1186 Complete_Tree) builds the tree one level at a
//  time: count, for every node of the level, the 4 x 4 (context base, predicted base) tables of the windows that
//  reach it; pick the context position with the most mutual information; set the node's probabilities, interpolated
//  with the parent's when the node saw fewer than SAMPLE_SIZE_BOUND windows.  Here the counting -- the only part
//  whose cost grows with the training set -- is one gmg_trainer_level_counts call per level (csrc/gmg_train.hip);
//  what is left is independent per node and runs on host threads, in the reference's own arithmetic (doubles, libm
//  log / logf) so that the written .icm is byte-identical.

#include "icm.hh"
#include "../../include/gmg.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

using namespace std;

namespace {

//  Get_Mutual_Info (src/ICM/icm.cc:1900-1955) of one 4 x 4 table whose entries sum to  sum .
double  Pair_Table_Info  (const int32_t * ct, int sum)
  {
   if  (sum == 0)
       return  0.0;

   double  left [ALPHABET_SIZE] = {0.0}, right [ALPHABET_SIZE] = {0.0};
   for  (int k = 0;  k < ALPHA_SQUARED;  k ++)
     {
      left [k / ALPHABET_SIZE] += ct [k];
      right [k % ALPHABET_SIZE] += ct [k];
     }
   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     {
      left [i] /= sum;
      right [i] /= sum;
     }

   double  info = 0.0;
   for  (int k = 0;  k < ALPHA_SQUARED;  k ++)
     {
      double  p = double (ct [k]) / sum;
      double  l = left [k / ALPHABET_SIZE], r = right [k % ALPHABET_SIZE];
      if  (p != 0.0 && l != 0.0 && r != 0.0)
          info += p * log (p / (l * r));
     }
   return  info;
  }


//  What one node learns from its tables ( npos  tables of 16 counts).
struct  Node_Choice_t
  {
   int  max_pos;          // context position chosen (before the not-enough-information rule)
   double  best_info;     // the largest mutual information seen
   double  used_info;     // the mutual information of the chosen position
   int  sum;              // windows that reached the node
   int  final_char_ct [ALPHABET_SIZE];   // of them, by predicted base
  };

//  The scan of src/ICM/icm.cc:1105-1139 (and :1393-1426 for a root): a position further right wins when its
//  information is within MUT_INFO_BIAS of the best so far.
void  Choose_Position  (const int32_t * ct, int npos, Node_Choice_t & c)
  {
   c . sum = 0;
   for  (int j = 0;  j < ALPHABET_SIZE;  j ++)
     c . final_char_ct [j] = 0;
   for  (int k = 0;  k < ALPHA_SQUARED;  k ++)
     {
      c . sum += ct [k];
      c . final_char_ct [k % ALPHABET_SIZE] += ct [k];
     }

   c . max_pos = 0;
   c . best_info = c . used_info = Pair_Table_Info (ct, c . sum);
   for  (int i = 1;  i < npos;  i ++)
     {
      double  next_info = Pair_Table_Info (ct + ALPHA_SQUARED * i, c . sum);
      if  (next_info >= c . best_info)
          {
           c . used_info = c . best_info = next_info;
           c . max_pos = i;
          }
      else if  (next_info >= (c . best_info / (1.0 + MUT_INFO_BIAS)))
          {
           c . max_pos = i;
           c . used_info = next_info;
          }
     }
  }


//  Interpolate_Probs (src/ICM/icm.cc:1260-1330):  prob  = the node's row,  parent  = its parent's (both plain
//  probabilities at this point).  Each store to a float rounds, as in the reference.
void  Blend_With_Parent  (float * prob, const float * parent, const int * ct)
  {
   double  total_sum = 0.0;
   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     total_sum += ct [i];

   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     prob [i] = (ct [i] + PSEUDO_COUNT * parent [i]) / (total_sum + PSEUDO_COUNT);
   if  (total_sum >= SAMPLE_SIZE_BOUND)
       return;

   double  chi2_stat = 0.0;
   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     {
      double  expected = total_sum * parent [i];
      if  (expected > 0.0)
          chi2_stat += pow (ct [i] - expected, 2.0) / expected;
     }

   int  e = 0;
   while  (e < NUM_CHI2_ENTRIES && CHI2_VAL [e] < chi2_stat)
     e ++;

   double  lambda;
   if  (e == 0)
       lambda = 0.0;
   else if  (e == NUM_CHI2_ENTRIES)
       lambda = 1.0;
     else   // the differences of table entries are float subtractions in the reference; keep them so
       lambda = CHI2_SIGNIFICANCE [e - 1]
                  + ((chi2_stat - CHI2_VAL [e - 1]) / (CHI2_VAL [e] - CHI2_VAL [e - 1]))
                      * (CHI2_SIGNIFICANCE [e] - CHI2_SIGNIFICANCE [e - 1]);

   lambda *= total_sum / SAMPLE_SIZE_BOUND;
   if  (lambda > 1.0)
       lambda = 1.0;

   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
     {
      prob [i] *= lambda;
      prob [i] += (1.0 - lambda) * parent [i];
     }
  }


int  First_Node_Of_Level  (int level)
  {
   int  pw = 1;
   for  (int i = 0;  i < level;  i ++)
     pw *= ALPHABET_SIZE;
   return  (pw - 1) / (ALPHABET_SIZE - 1);
  }


//  A few host threads for the length of one training: the nodes of a level do not depend on each other, the pieces of
//  the packed strings neither.  Run (n, body) calls body (i) for every i in [0, n), handing out  grain  of them at a
//  time (the tables of a level are unevenly filled), the caller working along; small jobs run on the caller alone.
class  Host_Workers_t
  {
  private:
   vector <thread>  threads;
   mutex  m;
   condition_variable  wake, all_done;
   const function <void (int)>  * job;
   int  job_n, job_grain;
   atomic <int>  next;
   int  generation, busy;
   bool  stop;

   void  Drain  (void)
     {
      for  ( ; ; )
        {
         int  lo = next . fetch_add (job_grain);
         if  (lo >= job_n)
             break;
         int  hi = (lo + job_grain < job_n ? lo + job_grain : job_n);
         for  (int i = lo;  i < hi;  i ++)
           (* job) (i);
        }
     }
   void  Loop  (void)
     {
      int  seen = 0;
      unique_lock <mutex>  lk (m);
      for  ( ; ; )
        {
         wake . wait (lk, [&] { return  stop || generation != seen; });
         if  (stop)
             return;
         seen = generation;
         lk . unlock ();
         Drain ();
         lk . lock ();
         if  (-- busy == 0)
             all_done . notify_one ();
        }
     }

  public:
   Host_Workers_t  ()  :  job (NULL), job_n (0), job_grain (1), next (0), generation (0), busy (0), stop (false)
     {
      unsigned  hw = thread :: hardware_concurrency ();
      int  workers = int (hw == 0 ? 1 : (hw > 16 ? 16 : hw));
      for  (int w = 1;  w < workers;  w ++)
        threads . push_back (thread (& Host_Workers_t :: Loop, this));
     }
   ~ Host_Workers_t  ()
     {
      {
       lock_guard <mutex>  g (m);
       stop = true;
      }
      wake . notify_all ();
      for  (size_t w = 0;  w < threads . size ();  w ++)
        threads [w] . join ();
     }
   void  Run  (int n, const function <void (int)> & body, int serial_below, int grain)
     {
      if  (n < serial_below || threads . empty ())
          {
           for  (int i = 0;  i < n;  i ++)
             body (i);
           return;
          }
      {
       lock_guard <mutex>  g (m);
       job = & body;
       job_n = n;
       job_grain = (grain > 0 ? grain : 1);
       next = 0;
       busy = int (threads . size ());
       generation ++;
      }
      wake . notify_all ();
      Drain ();
      unique_lock <mutex>  lk (m);
      all_done . wait (lk, [&] { return  busy == 0; });
     }
  };


//  One page-locked host buffer for the count tables, kept for the life of the process: a program that trains many
//  models (one per genome, src/../scripts/train_all.py style, through gmg_icm_train) pays the page-locking once.
//  A second training running at the same time gets an ordinary buffer of its own.
struct  Table_Buffer_t
  {
   int32_t  * p;
   size_t  ints;
   bool  cached;

   static mutex  & Lock  (void)  { static mutex  m;  return  m; }
   static int32_t  * & Cached_P  (void)  { static int32_t  * p = NULL;  return  p; }
   static size_t  & Cached_Ints  (void)  { static size_t  n = 0;  return  n; }
   static bool  & Cached_Busy  (void)  { static bool  b = false;  return  b; }

   explicit  Table_Buffer_t  (size_t n)  :  p (NULL), ints (n), cached (false)
     {
      {
       lock_guard <mutex>  g (Lock ());
       if  (! Cached_Busy ())
           {
            if  (Cached_Ints () < n)
                {
                 if  (Cached_P () != NULL)
                     {
                      gmg_host_unregister (Cached_P ());
                      free (Cached_P ());
                     }
                 Cached_P () = (int32_t *) calloc (n, sizeof (int32_t));
                 Cached_Ints () = (Cached_P () != NULL ? n : 0);
                 if  (Cached_P () != NULL)
                     gmg_host_register (Cached_P (), n * sizeof (int32_t));   // a refusal only costs speed
                }
            if  (Cached_P () != NULL)
                {
                 Cached_Busy () = cached = true;
                 p = Cached_P ();
                }
           }
      }
      if  (p == NULL)
          p = (int32_t *) calloc (n ? n : 1, sizeof (int32_t));
     }
   ~ Table_Buffer_t  ()
     {
      if  (cached)
          {
           lock_guard <mutex>  g (Lock ());
           Cached_Busy () = false;
          }
        else
          free (p);
     }
  };

}  // namespace



ICM_Training_t :: ICM_Training_t
    (int w, int d, int p)  :  ICM_t (w, d, p)
  {
  }


ICM_Training_t :: ~ ICM_Training_t
    ()
  {
  }


void  ICM_Training_t :: Train_Model
    (const vector <char *> & data)

//  src/ICM/icm.cc:1356-1455.  A failure of the device layer is fatal, as every other error of this class.

  {
   string  err;
   if  (! Try_Train_Model (data . empty () ? NULL : & data [0], int (data . size ()), err))
       {
        fprintf (stderr, "ERROR:  %s\n", err . c_str ());
        exit (EXIT_FAILURE);
       }
  }


bool  ICM_Training_t :: Try_Train_Model
    (const char * const * data, int string_ct, string & err)
  {
   const int  npos = (model_len > 1 ? model_len - 1 : 1);
   gmg_reads  * strings = NULL;
   gmg_trainer  * trainer = NULL;
   bool  ok = false;

   Invalidate_Device_Mirror ();

   //  GMG_TRAIN_TIMING=1: wall time of every stage on stderr
   long long  timing_opt = 0;
   gmg_get_option ("train_timing", & timing_opt);
   const bool  timing = (timing_opt != 0);
   chrono :: steady_clock :: time_point  t_prev = chrono :: steady_clock :: now ();
   auto  lap = [&] (const char * what, int level)
     {
      if  (! timing)
          return;
      chrono :: steady_clock :: time_point  t = chrono :: steady_clock :: now ();
      fprintf (stderr, "[gmg_train] %-10s level %2d %9.3f ms\n", what, level,
               chrono :: duration <double, milli> (t - t_prev) . count ());
      t_prev = t;
     };

   {
    const char  * env = getenv ("GMG_DEVICE");
    if  (gmg_init (env ? atoi (env) : 0) != GMG_OK)
        {
         err = string ("ICM_Training_t::Train_Model: ") + gmg_last_error ();
         return  false;
        }
   }

   //  the training strings as one packed batch in HBM; characters become codes exactly as Subscript maps them
   Host_Workers_t  workers;
   vector <uint64_t>  off (string_ct + 1, 0);
   {
    //  the lengths on all workers (64 MB of strings take 3 ms to measure on one), then their running sum
    uint64_t  * len = off . data () + 1;
    workers . Run ((string_ct + 255) / 256, [=] (int c)
      {
       const int  hi = (256 * (c + 1) < string_ct ? 256 * (c + 1) : string_ct);
       for  (int i = 256 * c;  i < hi;  i ++)
         len [i] = strlen (data [i]);
      }, 8, 1);
    for  (int i = 0;  i < string_ct;  i ++)
      off [i + 1] += off [i];
   }
   //  zero pages from calloc: each is first touched by the worker that packs into it
   struct  Words_t
     {
      uint32_t  * p;
      explicit  Words_t  (size_t n)  :  p ((uint32_t *) calloc (n ? n : 1, sizeof (uint32_t)))  {}
      ~ Words_t  ()  { free (p); }
      uint32_t  * data  (void)  { return  p; }
     }  packed (gmg_packed_words (off [string_ct]));
   Table_Buffer_t  counts (size_t (periodicity) * (First_Node_Of_Level (model_depth + 1) - First_Node_Of_Level (model_depth))
                             * npos * ALPHA_SQUARED);
   vector <int16_t>  mip_prev;
   if  (packed . p == NULL || counts . p == NULL)
       {
        err = "ICM_Training_t::Train_Model: out of host memory";
        return  false;
       }

   {
    //  pieces of 2^20 bases (a multiple of the 16 bases of a packed word, so no two threads share a word)
    const uint64_t  piece = 1 << 20, total = off [string_ct];
    const uint64_t  * offp = off . data ();
    uint32_t  * words = packed . data ();
    workers . Run (int ((total + piece - 1) / piece), [=] (int c)
      {
       const uint64_t  lo = piece * c, hi = (lo + piece < total ? lo + piece : total);
       int  s = int (upper_bound (offp, offp + string_ct + 1, lo) - offp) - 1;    // string holding base lo
       for  ( ;  s < string_ct && offp [s] < hi;  s ++)
         {
          const uint64_t  b = (offp [s] > lo ? offp [s] : lo), e = (offp [s + 1] < hi ? offp [s + 1] : hi);
          if  (e > b)
              gmg_pack_bases (data [s] + (b - offp [s]), e - b, b, words);
         }
      }, 2, 1);
   }
   if  (gmg_reads_upload (packed . data (), off . data (), string_ct, & strings) != GMG_OK
          || gmg_trainer_create (strings, model_len, model_depth, periodicity, & trainer) != GMG_OK)
       goto  Fail;
   lap ("upload", -1);

   //  the tables of a level land in a page-locked buffer: the last level's copy back (35 MB for the default shape)
   //  otherwise runs at pageable-memory speed and costs more than its counting
   lap ("pin", -1);

   for  (int level = 0;  level <= model_depth;  level ++)
     {
      const int  first = First_Node_Of_Level (level);
      const int  on_level = First_Node_Of_Level (level + 1) - first;

      if  (level > 0)
          {
           //  the device takes one descent step per level: it needs the positions the level above chose
           const int  pfirst = First_Node_Of_Level (level - 1), pon = on_level / ALPHABET_SIZE;
           mip_prev . resize (size_t (periodicity) * pon);
           for  (int f = 0;  f < periodicity;  f ++)
             for  (int k = 0;  k < pon;  k ++)
               mip_prev [size_t (f) * pon + k] = score [f] [pfirst + k] . mut_info_pos;
          }
      if  (gmg_trainer_level_counts (trainer, level, level > 0 ? mip_prev . data () : NULL, counts . p)
             != GMG_OK)
          goto  Fail;
      lap ("counts", level);

      const int32_t  * all = counts . p;
      ICM_Score_Node_t  * * sc = score;
      const int  W = model_len, D = model_depth;

      workers . Run (periodicity * on_level, [=] (int idx)
        {
         const int  frame = idx / on_level, sub = first + idx % on_level;
         const int32_t  * ct = all + size_t (idx) * npos * ALPHA_SQUARED;
         ICM_Score_Node_t  & node = sc [frame] [sub];
         Node_Choice_t  c;

         if  (level == 0)
             {
              if  (D == 0)
                  {
                   //  src/ICM/icm.cc:1376-1389: only the predicted base is counted (Count_Single_Chars); with a
                   //  context the device counts pairs and table 0 holds the same totals, without one it counts
                   //  the predicted base into table 0's first row
                   int  ch_ct [ALPHABET_SIZE] = {0}, sum = 0;
                   for  (int k = 0;  k < ALPHA_SQUARED;  k ++)
                     ch_ct [k % ALPHABET_SIZE] += ct [k];
                   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
                     sum += ch_ct [i];
                   for  (int i = 0;  i < ALPHABET_SIZE;  i ++)
                     node . prob [i] = (ch_ct [i] + float (PSEUDO_COUNT / ALPHABET_SIZE)) / (sum + PSEUDO_COUNT);
                   node . mut_info_pos = -1;
                   return;
                  }
              //  src/ICM/icm.cc:1391-1432: the root's probabilities are float arithmetic as written there
              Choose_Position (ct, W - 1, c);
              for  (int j = 0;  j < ALPHABET_SIZE;  j ++)
                node . prob [j] = (c . final_char_ct [j] + float (PSEUDO_COUNT / ALPHABET_SIZE))
                                    / float (c . sum + PSEUDO_COUNT);
              node . mut_info_pos = (short int) c . max_pos;
              node . mut_info = float (c . best_info);
              return;
             }

         //  src/ICM/icm.cc:1085-1160
         if  (sc [frame] [PARENT (sub)] . mut_info_pos < 0)
             {
              node . mut_info_pos = -2;      // the parent stopped
              return;
             }
         Choose_Position (ct, W - 1, c);
         if  (c . best_info <= MUT_INFO_EPSILON && c . sum < SAMPLE_SIZE_BOUND)
             c . max_pos = -1;               // not enough information gain: the tree ends here
         node . mut_info_pos = (short int) c . max_pos;
         node . mut_info = float (c . used_info);
         Blend_With_Parent (node . prob, sc [frame] [PARENT (sub)] . prob, c . final_char_ct);
        }, 128, 16);
      lap ("nodes", level);
     }

   //  Take_Logs (src/ICM/icm.cc:1334-1352).  The argument is a float, so the reference's  log  is the float
   //  overload.
   {
    ICM_Score_Node_t  * * sc = score;
    const int  N = num_nodes, slice = 4096, per_frame = (N + slice - 1) / slice;
    workers . Run (periodicity * per_frame, [=] (int idx)
      {
       const int  f = idx / per_frame, lo = (idx % per_frame) * slice, hi = (lo + slice < N ? lo + slice : N);
       for  (int i = lo;  i < hi;  i ++)
         for  (int j = 0;  j < ALPHABET_SIZE;  j ++)
           {
            float  & p = sc [f] [i] . prob [j];
            p = (p > 0.0 ? logf (p) : - FLT_MAX);
           }
      }, 4, 1);
   }
   ok = true;
   lap ("logs", -1);

  Fail:
   if  (! ok)
       err = string ("ICM_Training_t::Train_Model: ") + gmg_last_error ();
   gmg_trainer_free (trainer);
   gmg_reads_free (strings);
   Invalidate_Device_Mirror ();
   return  ok;
  }
